"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid) -> stdout (markdown-ish)."""
import csv, collections, statistics, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    key = (r['Kernel_Name'][:56], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e3:.1f} ms over {len(rows)} dispatches")
print(f"{'kernel':56s} {'blocks':>14s} {'calls':>7s} {'avg_us':>9s} {'med_us':>9s} {'total_ms':>9s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{k[0]:56s} {str(k[1:]):>14s} {len(v):7d} {sum(v)/len(v):9.1f} {statistics.median(v):9.1f} {sum(v)/1e3:9.1f} {100*sum(v)/tot:6.1f}")
