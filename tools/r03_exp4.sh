#!/bin/bash
# round 3, call 4: the tuner on the measured pedigree set (candidates pre-built), the corrected LDS-DMA shell, the CLI's whole-process rate
set -u
O=$PWD/gpurun_out/r03e; mkdir -p $O
timeout -k 10 900 python tools/make_tuned_picks.py $O/tuned_picks.json > $O/make_tuned_picks.log 2>&1; echo "picks rc=$?"; tail -3 $O/make_tuned_picks.log
make -s tools/io_ceiling 2>&1 | tail -2
timeout -k 10 200 ./tools/io_ceiling 10 10000000 > $O/io_ceiling_ped10_10M.txt 2>&1; echo "io_ceiling rc=$?"; grep -i "dma\|ew16nt\|shell2\|differ" $O/io_ceiling_ped10_10M.txt | head -20
python tools/cli_throughput.py 3000000 > $O/cli_3M.txt 2>&1; grep "sites/s\|loop" $O/cli_3M.txt | head -12
