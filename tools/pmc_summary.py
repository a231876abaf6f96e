import csv, glob, collections, sys, json
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/runc/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'bn_enum' in r['Kernel_Name'] or 'bn_' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
        print("%-28s n=%d mean=%.6g" % (k, len(v), out[k]))
json.dump(out, open(sys.argv[1] + "/pmc_summary.json", "w"), indent=1)
