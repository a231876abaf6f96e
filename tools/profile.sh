#!/bin/bash
# tools/profile.sh TAG [bench args...] — rocprofv3 evidence for one bench configuration, on the GPU box.
#   kernel-trace + stats of the full-size run, then PMC passes (separate, as gfx950's TCC slots
#   require; never combined with API tracing) at 1 M sites.  Results land in gpurun_out/TAG/;
#   tools/pmc_summary.py condenses them; copy what is to be judged into profiles/.
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
# the library refuses to spawn hipcc under a profiler: make sure this run's kernels are in the cache first
python3 $R/bench.py --warm-only "$@" > $O/warm.log 2>&1 || { echo "warm-up failed"; cat $O/warm.log; exit 1; }
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/kt.log 2>&1
echo "kernel-trace rc=$?"
for c in FETCH_SIZE WRITE_SIZE \
  "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY" \
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" \
  "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $c | cut -d" " -f1)
  timeout -k 10 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --no-cpu-baseline --sites 1000000 --steps 2 --warmup 1 "$@" > $O/pmc_$tag.log 2>&1
  echo "$tag rc=$?"
done
cd $R
python3 tools/pmc_summary.py $O
cp $O/kt/*/*_kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
head -3 $O/kernel_stats.csv | cut -c1-220
# per-launch durations of our kernels (told apart by LDS size), then drop the raw traces: gpurun brings back 64 MiB at most
python3 - "$O" <<'PY'
import collections, csv, glob, json, sys
o = sys.argv[1]
per = collections.defaultdict(list)
for f in glob.glob(o + "/kt/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "famseq_" in r["Kernel_Name"] or "bn_enum" in r["Kernel_Name"]:
            per["%s[lds=%s]" % (r["Kernel_Name"].split("(")[0][:24], r["LDS_Block_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out = {k: {"launches": len(v), "ms": [round(x, 4) for x in v[:12]], "mean_ms_excluding_first": sum(v[1:]) / max(1, len(v) - 1)} for k, v in per.items()}
json.dump({"per_launch": out}, open(o + "/kernel_trace_durations.json", "w"), indent=1)
for k, v in out.items():
    print(k, v["launches"], "launches, mean %.4f ms" % v["mean_ms_excluding_first"])
PY
rm -rf $O/kt $O/pmc_*/
