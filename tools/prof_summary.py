"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table.
usage: python tools/prof_summary.py <results.db> <n_steps_in_trace> "<header line>" > profiles/<name>.txt"""
import re
import sqlite3
import sys

db, nsteps, header = sys.argv[1], int(sys.argv[2]), sys.argv[3]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("#", header)
print(f"# total kernel time {tot / 1e6:.1f} ms over {nsteps} steps -> {tot / nsteps / 1e6:.2f} ms per step")
print("# pct  calls  avg_us  min_us  max_us  name")
for r in rows[:40]:
    n = re.sub(r"\(anonymous namespace\)::", "", r[0])[:110]
    print(f"{100 * r[2] / tot:6.2f} {r[1]:6d} {r[3] / 1e3:9.1f} {r[4] / 1e3:9.1f} {r[5] / 1e3:9.1f}  {n}")
