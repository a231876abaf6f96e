#!/bin/bash
# round 3 on the GPU box: the GPU suite, smoke, the bench line, then the rocprofv3 evidence (TAG under gpurun_out/)
set -u
TAG=${TAG:-r03b}
mkdir -p gpurun_out/$TAG
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest_gpu.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/$TAG/smoke.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/$TAG/bench_ped10.json 2> gpurun_out/$TAG/bench_ped10.err; echo "bench rc=$?"
python - <<'PY'
import json,os
t=os.environ.get("TAG","r03b")
d=json.loads(open("gpurun_out/%s/bench_ped10.json"%t).read().strip().splitlines()[-1])
print("value %.4g sites/s, fp64 frac %.3f, kernel_ms %.3f"%(d["value"],d["fp64_valu"]["frac"],d["roofline"]["kernel_ms"]))
for k in ("elim_engine","configs_1_ped5","configs_4_ped15","elim_N32"):
    if k in d: print(k, "%.4g"%d[k]["value"], "frac %.3f"%d[k]["roofline"]["frac"], d[k].get("fp64_valu_frac"), d[k].get("outputs_valid"))
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
