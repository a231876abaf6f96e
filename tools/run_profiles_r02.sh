bash tools/profile.sh r02b > gpurun_out/profile_r02b.log 2>&1; tail -3 gpurun_out/profile_r02b.log
bash tools/traffic.sh r02b_traffic ped10 2>&1 | tail -2
bash tools/traffic.sh r02b_traffic5 ped5 2>&1 | tail -2
python tools/small_batch_rates.py ped10 > gpurun_out/small_batch_rates_ped10.txt 2>&1; tail -8 gpurun_out/small_batch_rates_ped10.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02b_call -- python3 $GRAFT_REPO_ROOT/tools/io_kernel_rates.py > $GRAFT_REPO_ROOT/gpurun_out/r02b_call.log 2>&1; cat $GRAFT_REPO_ROOT/gpurun_out/r02b_call/*/*kernel_stats.csv | cut -c1-140
cd $GRAFT_REPO_ROOT && cp gpurun_out/r02b_call/*/*kernel_stats.csv gpurun_out/r02b_call_kernel_stats.csv; rm -rf gpurun_out/r02b_call
du -sh gpurun_out
