"""Print per-kernel PMC counter sums from a rocprofv3 rocpd sqlite database.  usage: python tools/pmc_summary.py <db> [kernel substring]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
try:
    rows = c.execute(q).fetchall()
except Exception as e:
    print(cols); raise
for k, n, v, nd in rows:
    if sub in k:
        print(f"{re.sub(r'[(]anonymous namespace[)]::','',k)[:60]:60s} {n:28s} {v/nd:16.1f}  (per dispatch, {nd} dispatches)")
