#!/bin/bash
out=gpurun_out/exp/skel.txt
: > $out
run() { label=$1; shift
  line=$(env "$@" 2>>gpurun_out/exp/skel.err | tail -1)
  python3 - "$label" "$line" >> $out <<'PY'
import sys, json
d = json.loads(sys.argv[2]); r = d["roofline"]
print(sys.argv[1], "sites/s=%.4g" % d["value"], "kernel_ms=%.4f" % r["kernel_ms"], "hbm_frac=%.3f" % r["frac"])
PY
}
B="timeout -k 10 200 python3 bench.py --sites 4000000 --no-cpu-baseline --no-side-configs --steps 20 --warmup 3"
for lc in 0 1; do
  run "elim ped5 lc=$lc" A=1 $B --workload ped5 --engine elim --lc $lc || exit 1
  run "lane ped5 lc=$lc" A=1 $B --workload ped5 --no-elim --lc $lc || exit 1
  run "elim ped10 lc=$lc" A=1 $B --workload ped10 --engine elim --lc $lc || exit 1
  run "lane ped10 lc=$lc" A=1 $B --workload ped10 --no-elim --lc $lc --steps 3 || exit 1
done
cat $out
