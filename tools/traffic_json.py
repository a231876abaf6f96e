"""tools/traffic_json.py DIR WORKLOAD [ROUND] [--engine elim] — turn the two PMC passes of tools/traffic.sh
(DIR/pmc_summary.json: FETCH_SIZE and WRITE_SIZE means per launch, KiB) into profiles/hbm_traffic_<workload>.json,
the file bench.py reads for roofline.traffic.  The file is stamped with the content hash of the code object
the counters were measured on (the generated kernel's .hsaco name), so that bench.py quotes it only while it
still runs that code.  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md,
calibrated in profiles/r01a)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import famseq_amd as fs  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
d, workload = args[0], args[1]
rnd = args[2] if len(args) > 2 else os.path.basename(d.rstrip("/"))
elim = "--engine" in sys.argv and sys.argv[sys.argv.index("--engine") + 1] == "elim"
summ = json.load(open(os.path.join(d, "pmc_summary.json")))
kernel = "famseq_elim" if elim else "famseq_enum_lane"
fetch = [v for k, v in summ.items() if k.startswith(kernel) and k.endswith(":FETCH_SIZE")]
write = [v for k, v in summ.items() if k.startswith(kernel) and k.endswith(":WRITE_SIZE")]
assert len(fetch) == 1 and len(write) == 1, (fetch, write, list(summ))
ped = fs.synthetic_pedigree(workload)
ctx = fs.Context(fs.make_model(ped), device=-1)
ctx.set_option("enum_impl", 1)
if elim:
    ctx.set_option("engine", fs.ENGINE_ELIM)
plan = ctx.plan()
ctx.close()
code = os.path.basename(plan["elim_code_object" if elim else "enum_lane_code_object"]).split(".")[0]
sites = 1_000_000
out = {
    "workload": workload, "kernel": kernel, "kernel_hash": code, "sites_per_launch": sites,
    "fetch_bytes": fetch[0] * 1024 * 2, "write_bytes": write[0] * 1024,
    "bytes_per_launch": fetch[0] * 1024 * 2 + write[0] * 1024,
    "algorithmic_bytes_per_launch": sites * (72 * ped.n + 2),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/traffic.sh); FETCH_SIZE x2 on gfx950 "
              "(MI355X_MICROARCH.md; calibrated with tools/calib_fetch, profiles/r01a/calib_fetch_write.txt); counters are KiB",
    "round": rnd,
}
out["ratio_to_algorithmic"] = out["bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
# next to the counter summary (what gpurun brings back) and, when run in the repository itself, in profiles/
name = "hbm_traffic_%s%s.json" % (workload, "_elim" if elim else "")
json.dump(out, open(os.path.join(d, name), "w"), indent=1)
path = os.path.join(ROOT, "profiles", name)
json.dump(out, open(path, "w"), indent=1)
print(path, json.dumps(out))
