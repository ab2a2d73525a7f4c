"""Per-kernel statistics from a rocprofv3 results .db (sqlite): calls, total / average / min / max duration in us."""
import sqlite3, sys
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# {path}")
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct")
    for name, n, t, mn, mx in rows:
        name = name.split("(")[0]
        print(f"\"{name}\",{n},{t/1e3:.1f},{t/n/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*t/tot:.1f}")
