#!/bin/bash
# tools/traffic.sh TAG WORKLOAD — HBM bytes per launch of the enumeration kernel (FETCH_SIZE and
# WRITE_SIZE in separate passes, as gfx950 requires), 1 M sites.  Output: gpurun_out/TAG/…
set -u
TAG=$1; WL=$2
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
# the library refuses to spawn hipcc under a profiler: make sure this run's kernels are in the cache first
python3 $R/bench.py --warm-only --workload $WL > $O/warm.log 2>&1 || { echo "warm-up failed"; cat $O/warm.log; exit 1; }
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-side-configs --no-elim --sites 1000000 --steps 2 --warmup 1 > $O/pmc_$c.log 2>&1
  echo "$WL $c rc=$?"
done
cd $R
python3 tools/pmc_summary.py $O | grep "FETCH\|WRITE"
python3 tools/traffic_json.py $O $WL
rm -rf $O/pmc_*/
