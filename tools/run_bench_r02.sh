TAG=${TAG:-r02c}   # the round's tag: results under gpurun_out/$TAG*, to be copied into profiles/$TAG
# the bench lines kept under profiles/${TAG} (run on the GPU box)
mkdir -p gpurun_out/${TAG}_bench
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench/bench_ped10.json 2> gpurun_out/${TAG}_bench/bench_ped10.err
for w in ped5 ped15 trio quad; do python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench/bench_$w.json 2>/dev/null; done
python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --sites 2000000 --no-cpu-baseline > gpurun_out/${TAG}_bench/bench_2ranks_shared_gpu_weak.json 2>/dev/null
python bench.py --gpus 2 --share-gpu --steps 5 --warmup 2 --scaling strong --no-cpu-baseline > gpurun_out/${TAG}_bench/bench_2ranks_shared_gpu_strong.json 2>/dev/null
python tools/cli_throughput.py 3000000 --packed > gpurun_out/${TAG}_bench/cli_packed_3M.txt 2>&1
python tools/cli_throughput.py 12000000 --packed > gpurun_out/${TAG}_bench/cli_packed_12M.txt 2>&1
tail -5 gpurun_out/${TAG}_bench/cli_packed_12M.txt
python tools/host_path_rate.py > gpurun_out/${TAG}_bench/host_path_rate.txt 2>&1; tail -6 gpurun_out/${TAG}_bench/host_path_rate.txt
ls -la gpurun_out/${TAG}_bench
