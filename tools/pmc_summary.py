"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, collections, sys, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r['Kernel_Name'][:60], r['Grid_Size'])
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
print(f"{'kernel':60s} {'grid':>10s} {'n':>5s}  counters (mean per dispatch)")
for k, v in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = len(next(iter(v.values())))
    print(f"{k[0]:60s} {k[1]:>10s} {n:5d}  " + "  ".join(f"{c}={sum(x)/len(x):.4g}" for c, x in v.items()))
