"""tools/pmc_summary.py DIR — condense rocprofv3 --pmc passes (DIR/pmc_*/…/*_counter_collection.csv)
into DIR/pmc_summary.json: mean counter value per launch, keyed by kernel and its LDS block size
(one bench run launches the same generated entry point for more than one pedigree)."""
import collections
import csv
import glob
import json
import sys

out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bn_enum" in r["Kernel_Name"] or "famseq_" in r["Kernel_Name"] or "_kernel" in r["Kernel_Name"][:40] and "pl16" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].split("::")[-1][:24]
            agg[("%s[lds=%s]" % (name, r["LDS_Block_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out["%s:%s" % k] = sum(v) / len(v)
        print("%-34s %-26s n=%d mean=%.6g" % (k[0], k[1], len(v), out["%s:%s" % k]))
json.dump(out, open(sys.argv[1] + "/pmc_summary.json", "w"), indent=1)
