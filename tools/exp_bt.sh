#!/bin/bash
# usage: exp_bt.sh  -> sweeps block sizes of the generated kernels
out=gpurun_out/exp/bt_sweep.txt
: > $out
run() { # label, env..., -- args
  label=$1; shift
  line=$(env "$@" 2>>gpurun_out/exp/bt_sweep.err | tail -1)
  python3 - "$label" "$line" >> $out <<'PY'
import sys, json
d = json.loads(sys.argv[2])
r = d["roofline"]
print(sys.argv[1], "sites/s=%.4g" % d["value"], "kernel_ms=%.4f" % r["kernel_ms"], "hbm_frac=%.3f" % r["frac"])
PY
}
for bt in 64 128 256 512; do
  run "elim ped10 BT=$bt" FAMSEQ_ELIM_BT=$bt timeout -k 10 200 python3 bench.py --workload ped10 --sites 4000000 --engine elim --no-cpu-baseline --no-side-configs --steps 10 --warmup 3 || exit 1
  run "elim ped5 BT=$bt" FAMSEQ_ELIM_BT=$bt timeout -k 10 200 python3 bench.py --workload ped5 --sites 4000000 --engine elim --no-cpu-baseline --no-side-configs --steps 20 --warmup 3 || exit 1
  run "lane ped5 BT=$bt" FAMSEQ_LANE_BT=$bt timeout -k 10 200 python3 bench.py --workload ped5 --sites 4000000 --no-elim --no-cpu-baseline --no-side-configs --steps 20 --warmup 3 || exit 1
done
cat $out
