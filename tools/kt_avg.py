"""average / min duration per kernel of a rocprofv3 --kernel-trace results.db.   usage: python tools/kt_avg.py <db> [substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
for name, n, avg, mn in db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name"):
    if sub in name and "at6native" not in name:
        print(f"{name[:60]:60s} n={n:4d} avg {avg / 1000:8.1f} us  min {mn / 1000:8.1f} us")
