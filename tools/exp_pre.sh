#!/bin/bash
out=gpurun_out/exp/pre.txt
: > $out
run() { label=$1; shift
  line=$(env "$@" 2>>gpurun_out/exp/pre.err | tail -1)
  python3 - "$label" "$line" >> $out <<'PY'
import sys, json
d = json.loads(sys.argv[2]); r = d["roofline"]
print(sys.argv[1], "sites/s=%.4g" % d["value"], "kernel_ms=%.4f" % r["kernel_ms"], "hbm_frac=%.3f" % r["frac"])
PY
}
B="timeout -k 10 200 python3 bench.py --sites 4000000 --no-cpu-baseline --no-side-configs --steps 20 --warmup 3"
for early in 0 6 10; do
  for mw in 2 3 4; do
    run "elim ped5 early<=$early MW=$mw" FAMSEQ_PREFETCH_EARLY_MAXN=$early FAMSEQ_ELIM_MINWAVES=$mw $B --workload ped5 --engine elim || exit 1
    run "lane ped5 early<=$early MW=$mw" FAMSEQ_PREFETCH_EARLY_MAXN=$early FAMSEQ_LANE_MINWAVES=$mw $B --workload ped5 --no-elim || exit 1
  done
  run "elim ped10 early<=$early" FAMSEQ_PREFETCH_EARLY_MAXN=$early $B --workload ped10 --engine elim || exit 1
done
cat $out
