"""Per-kernel summary of a rocprofv3 --kernel-trace results .db (sqlite): calls, average, total, share."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
tot = cur.execute(f"select sum(end-start)/1e6 from {kt}").fetchone()[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print(f"total kernel time {tot:.1f} ms")
print(f"{'kernel':100s} {'grid':>10s} {'calls':>7s} {'avg_us':>9s} {'total_ms':>9s} {'%':>5s}")
q = (f"select s.kernel_name, d.grid_size_x/d.workgroup_size_x, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1e6 from {kt} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x order by 5 desc limit {n}")
for r in cur.execute(q):
    print(f"{r[0][:100]:100s} {r[1]:10d} {r[2]:7d} {r[3]:9.1f} {r[4]:9.1f} {100*r[4]/tot:5.1f}")
