"""tools/make_tuned_picks.py OUT.json — on the GPU box: famseq_set_option(ctx, "tune", 1) for every pedigree build()
pre-builds kernels for (__graft_entry__.build_pedigrees), into a scratch kernel cache; the picks go to OUT.json, to be
committed as famseq_amd/tuned_picks.json, from which build() ships them as notes next to the pre-built code objects."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("FAMSEQ_PICKS_SCRATCH_CACHE"):  # (default: the in-tree cache, where tools/prebuild_candidates.py put the candidates)
    os.environ["FAMSEQ_KERNEL_CACHE"] = tempfile.mkdtemp(prefix="famseq_picks_")
import famseq_amd as fs  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

out = sys.argv[1]
peds = ge.tuned_pedigrees()
table, t0 = {}, time.time()
for k, ped in enumerate(peds):
    key = ge.pedigree_key(ped)
    if key in table or ped.n > fs.MAXN:  # (beyond 20 members there is one engine and no enumeration to race)
        continue
    ctx = fs.Context(fs.make_model(ped), enum_impl=1)
    ctx.set_option("tune", 1)
    plan = ctx.plan()
    entry = {"lane": plan["enum_lane_variant"], "elim": -1, "report": plan["tune"]}
    if plan["elim_supported"]:
        ctx.set_option("engine", fs.ENGINE_ELIM)
        entry["elim"] = ctx.plan()["elim_variant"]
    ctx.close()
    table[key] = entry
    print("%3d/%d n=%2d lane v%d elim v%d  [%.0f s]" % (k + 1, len(peds), ped.n, entry["lane"], entry["elim"], time.time() - t0), flush=True)
    json.dump(table, open(out, "w"), indent=0, sort_keys=True)
print("wrote", out, len(table), "pedigrees")
