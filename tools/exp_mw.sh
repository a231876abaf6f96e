#!/bin/bash
out=gpurun_out/exp/mw_sweep.txt
: > $out
run() { label=$1; shift
  line=$(env "$@" 2>>gpurun_out/exp/mw_sweep.err | tail -1)
  python3 - "$label" "$line" >> $out <<'PY'
import sys, json
d = json.loads(sys.argv[2]); r = d["roofline"]
print(sys.argv[1], "sites/s=%.4g" % d["value"], "kernel_ms=%.4f" % r["kernel_ms"], "hbm_frac=%.3f" % r["frac"])
PY
}
B="timeout -k 10 200 python3 bench.py --sites 4000000 --no-cpu-baseline --no-side-configs --steps 20 --warmup 3"
for mw in 2 3 4; do
  run "elim ped5 MW=$mw" FAMSEQ_ELIM_MINWAVES=$mw $B --workload ped5 --engine elim || exit 1
  run "elim ped5 MW=$mw nopre" FAMSEQ_ELIM_MINWAVES=$mw FAMSEQ_PREFETCH_MAXN=0 $B --workload ped5 --engine elim || exit 1
  run "lane ped5 MW=$mw" FAMSEQ_LANE_MINWAVES=$mw $B --workload ped5 --no-elim || exit 1
  run "lane ped5 MW=$mw nopre" FAMSEQ_LANE_MINWAVES=$mw FAMSEQ_PREFETCH_MAXN=0 $B --workload ped5 --no-elim || exit 1
  run "elim ped10 MW=$mw" FAMSEQ_ELIM_MINWAVES=$mw $B --workload ped10 --engine elim || exit 1
  run "elim ped10 MW=$mw nopre" FAMSEQ_ELIM_MINWAVES=$mw FAMSEQ_PREFETCH_MAXN=0 $B --workload ped10 --engine elim || exit 1
done
cat $out
