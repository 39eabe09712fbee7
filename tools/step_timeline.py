"""Wall-time attribution of ONE captured denoise step from a rocprofv3 kernel trace (rocpd sqlite): the step's span is cut at
every kernel start / end; each slice is shared equally by the kernels running in it (the low-resolution section runs on several
streams) or booked as `(gap)` when nothing runs.  Per kernel name: launches, summed duration, attributed wall time.
usage: python tools/step_timeline.py <results.db> [--seq]   (--seq: also print the launches of the step in start order)"""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n)[:100]
# a step ends with the fused CFG + DDIM kernel; take the last complete one
ends = [i for i, r in enumerate(rows) if "cfg_ddim" in r[0]]
if len(ends) < 2:
    sys.exit("need at least two steps in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
step = rows[lo:hi]
t0, t1 = step[0][1], max(r[2] for r in step)
ev = []
for i, (n, s, e) in enumerate(step):
    ev.append((s, 1, i))
    ev.append((e, 0, i))
ev.sort()
live = set()
wall = collections.defaultdict(float)
dur = collections.defaultdict(float)
cnt = collections.Counter()
prev = t0
conc = collections.defaultdict(float)
for t, kind, i in ev:
    if t > prev:
        if live:
            for j in live:
                wall[short(step[j][0])] += (t - prev) / len(live)
        else:
            wall["(gap)"] += t - prev
        conc[min(len(live), 4)] += t - prev
        prev = t
    if kind:
        live.add(i)
    else:
        live.discard(i)
for n, s, e in step:
    dur[short(n)] += e - s
    cnt[short(n)] += 1
print(f"# one step: {len(step)} launches, span {(t1 - t0) / 1e6:.3f} ms; kernels columns: {cols}")
print("# time with k kernels in flight (ms): " + ", ".join(f"{k}{'+' if k == 4 else ''}: {v / 1e6:.2f}" for k, v in sorted(conc.items())))
print("# wall_ms  wall%  launches  sum_dur_ms  avg_us  name")
for n, w in sorted(wall.items(), key=lambda kv: -kv[1])[:45]:
    k = cnt.get(n, 0)
    print(f"{w / 1e6:8.3f} {100 * w / (t1 - t0):6.2f} {k:6d} {dur.get(n, 0) / 1e6:9.3f} {(dur.get(n, 0) / k / 1e3 if k else 0):8.1f}  {n}")
if "--seq" in sys.argv:
    for n, s, e in step:
        print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {short(n)[:70]}")
