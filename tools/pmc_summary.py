import csv, glob, collections, sys, json
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/runc/*_counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if 'bn_enum' in r['Kernel_Name'] or 'famseq_' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0].split('::')[-1][:24], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k, v in agg.items():
        out["%s:%s" % k] = sum(v) / len(v)
        print("%-26s %-26s n=%d mean=%.6g" % (k[0], k[1], len(v), out["%s:%s" % k]))
json.dump(out, open(sys.argv[1] + "/pmc_summary.json", "w"), indent=1)
