"""Per-kernel SQ utilisation over one rocprofv3 --pmc pass of bench.py: MFMA-busy, VALU-active, parked and issue-stalled
fractions of the wave cycles.  usage: python tools/pmc_step_summary.py <db>"""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                 "group by kernel_name, counter_name").fetchall()
k = {}
for name, cn, v, nd in rows:
    k.setdefault(name, {})[cn] = v
    k[name]["_n"] = nd
tot = sum(d.get("SQ_WAVE_CYCLES", 0) for d in k.values())
print("# share of wave-cycles | calls | mfma_busy/busy | valu | lds | parked(WAIT_ANY) | issue-stall(WAIT_INST) | kernel")
for name, d in sorted(k.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:28]:
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    busy = d.get("SQ_BUSY_CYCLES", 0) or 1   # per-SE quad... used only for ordering
    f = lambda x: 100.0 * d.get(x, 0) / wc
    mf = 100.0 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 4.0 / wc  # cycles -> quad-cycles; relative to WAVE cycles (all waves)
    n = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
    print(f"{d['_n']:6d} busy={busy/1e6:9.1f}M  mfma/wave={mf:5.1f}%  valu={f('SQ_ACTIVE_INST_VALU'):5.1f}%  lds={f('SQ_ACTIVE_INST_LDS'):5.1f}%  "
          f"parked={f('SQ_WAIT_ANY'):5.1f}%  stall={f('SQ_WAIT_INST_ANY'):5.1f}%  {n}")
