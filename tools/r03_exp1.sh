#!/bin/bash
# round 3, experiment 1: scalar-table / prefetch variants of the lane kernel (tools/exp_variants) + parity under the knobs
set -u
mkdir -p gpurun_out/r03a
make -s tools/kernel_bench 2>&1 | tail -2
for rep in 1 2; do bash tools/exp_kb.sh; done > gpurun_out/r03a/kb.txt 2>&1
cat gpurun_out/r03a/kb.txt
FAMSEQ_LANE_ST=1 FAMSEQ_LANE_PRE=2 FAMSEQ_KERNEL_CACHE=/tmp/kc_st timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03a/parity_st.txt 2>&1
tail -5 gpurun_out/r03a/parity_st.txt
