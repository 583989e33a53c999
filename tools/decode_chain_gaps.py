"""Per kernel of the decode chain, from a rocprofv3 --kernel-trace CSV: average duration and the average idle time between its end and the start of the
next kernel of the chain (launch / dependency gap).  Only dispatches whose successor starts within 50 us are counted (i.e. inside a decode graph).
usage: python tools/decode_chain_gaps.py <rocprof output dir>"""
import csv, collections, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
chain = ("skinny", "decode_attn", "add_rmsnorm", "greedy")
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    if not any(c in a["Kernel_Name"] for c in chain) or not any(c in b["Kernel_Name"] for c in chain):
        continue
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    if g > 50:
        continue
    key = (a["Kernel_Name"][:60], int(a["Grid_Size_X"]) // int(a["Workgroup_Size_X"]), int(a["Grid_Size_Y"]))
    dur[key].append((int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3); gap[key].append(g)
print(f"{'kernel':60s} {'blocks':>12s} {'calls':>7s} {'avg_us':>8s} {'gap_after_us':>13s} {'sum_per_call':>13s}")
tot = 0.0
for k in sorted(dur, key=lambda k: -sum(dur[k]) - sum(gap[k])):
    n = len(dur[k]); d = sum(dur[k]) / n; g = sum(gap[k]) / n
    print(f"{k[0]:60s} {str(k[1:]):>12s} {n:7d} {d:8.2f} {g:13.2f} {d + g:13.2f}")
