#!/bin/bash
# round 3, experiment 3: registers-first sum-product kernels on small pedigrees; the GPU suite on the current tree
set -u
O=$PWD/gpurun_out/r03d; mkdir -p $O
make -s tools/kernel_bench 2>&1 | tail -2
SITES=4000000 bash tools/exp_kb.sh > $O/kb_elim_small.txt 2>&1; SITES=4000000 bash tools/exp_kb.sh >> $O/kb_elim_small.txt 2>&1
cat $O/kb_elim_small.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
python tools/small_batch_rates.py ped10 > $O/small_batch_rates_ped10.txt 2>&1; tail -12 $O/small_batch_rates_ped10.txt
