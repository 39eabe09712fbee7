"""LDS bank-conflict share per kernel of one denoise step: usage (GPU box)
   cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/ldsc -o p -- python $R/bench.py --steps 1 --warmup 1 --step-only; python tools/lds_conflicts.py <db>"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
d = {}
for k, n, v, nd in rows:
    d.setdefault(k, {})[n] = (v, nd)
out = []
for k, m in d.items():
    if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"][0] > 0:
        act, nd = m["SQ_LDS_IDX_ACTIVE"]
        conf = m.get("SQ_LDS_BANK_CONFLICT", (0, nd))[0]
        out.append((conf, act, nd, k))
print("# conflict cycles (M) | LDS cycles (M) | share | dispatches | kernel   (sums over the run: set-up + 3 steps)")
for conf, act, nd, k in sorted(out, reverse=True)[:30]:
    print(f"{conf / 1e6:10.1f} {act / 1e6:10.1f} {conf / act:6.1%} {nd:6d}  {re.sub(r'[(]anonymous namespace[)]::', '', k)[:100]}")
