#!/bin/bash
# round 3, experiment 2: sum-product kernel with the likelihoods in registers (tools/exp_variants el*), the table-driven
# Phred in the call path (rocprofv3 kernel trace), the wide-pedigree GPU tests
set -u
O=$PWD/gpurun_out/r03c; mkdir -p $O
make -s tools/kernel_bench 2>&1 | tail -2
SITES=4000000 bash tools/exp_kb.sh > $O/kb_elim.txt 2>&1; SITES=4000000 bash tools/exp_kb.sh >> $O/kb_elim.txt 2>&1
cat $O/kb_elim.txt
timeout -k 10 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_call.py tests/test_cli_gpu.py -m gpu -x -q > $O/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_subset.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/call -- python3 $R/tools/io_kernel_rates.py > $O/call.log 2>&1; echo "call rc=$?"
cd $R; cp $O/call/*/*kernel_stats.csv $O/call_path_kernel_stats.csv 2>/dev/null; rm -rf $O/call; cut -c1-150 $O/call_path_kernel_stats.csv | head -5
python tools/cli_throughput.py 3000000 > $O/cli_3M.txt 2>&1; tail -6 $O/cli_3M.txt
