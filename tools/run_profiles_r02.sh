TAG=${TAG:-r02c}   # the round's tag: results under gpurun_out/$TAG*, to be copied into profiles/$TAG
bash tools/profile.sh ${TAG} > gpurun_out/profile_${TAG}.log 2>&1; tail -3 gpurun_out/profile_${TAG}.log
bash tools/traffic.sh ${TAG}_traffic ped10 2>&1 | tail -2
bash tools/traffic.sh ${TAG}_traffic5 ped5 2>&1 | tail -2
python tools/small_batch_rates.py ped10 > gpurun_out/small_batch_rates_ped10.txt 2>&1; tail -8 gpurun_out/small_batch_rates_ped10.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_call -- python3 $GRAFT_REPO_ROOT/tools/io_kernel_rates.py > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_call.log 2>&1; cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_call/*/*kernel_stats.csv | cut -c1-140
cd $GRAFT_REPO_ROOT && cp gpurun_out/${TAG}_call/*/*kernel_stats.csv gpurun_out/${TAG}_call_kernel_stats.csv; rm -rf gpurun_out/${TAG}_call
du -sh gpurun_out
