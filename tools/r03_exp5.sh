#!/bin/bash
# round 3, call 5: the sum-product kernel without LDS staging (variants 8+) against the staged registers-first form, 24-96 members
set -u
O=$PWD/gpurun_out/r03g; mkdir -p $O
make -s tools/kernel_bench 2>&1 | tail -2
SITES=2000000 bash tools/exp_kb.sh > $O/kb_direct.txt 2>&1; SITES=2000000 bash tools/exp_kb.sh >> $O/kb_direct.txt 2>&1
cat $O/kb_direct.txt
