"""Per-launch HBM traffic of the roofline kernel from rocprofv3 PMC passes (see tools/pmc_traffic.sh).
usage: python tools/pmc_traffic_summary.py gpurun_out/traffic  -> prints JSON (copy into profiles/)"""
import glob, json, os, sqlite3, sys

root = sys.argv[1]


def per_dispatch(sub, counter, kernel):
    """list of per-dispatch counter values (summed over instances) for kernels whose name contains `kernel`"""
    dbs = glob.glob(os.path.join(root, sub, "**", "*.db"), recursive=True)
    if not dbs:
        return []
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select dispatch_id, sum(value) from counters_collection where counter_name = ? and kernel_name like ? "
                     "group by dispatch_id order by dispatch_id", (counter, "%" + kernel + "%")).fetchall()
    return [r[1] for r in rows]


out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; counter unit = KB", "calibration": {}}
# known byte counts: rbw kA reads 32.768 MB then 262.144 MB (contiguous 16 B/lane); wbw kA writes 131.072 MB
rd = per_dispatch("FETCH_SIZE_rbw", "FETCH_SIZE", "kA")
wr = per_dispatch("WRITE_SIZE_wbw", "WRITE_SIZE", "kA")
if rd:
    big = max(rd)
    out["calibration"]["read_probe_bytes"] = 262144000
    out["calibration"]["read_probe_FETCH_SIZE_KB"] = big
    out["calibration"]["read_factor"] = 262144000 / (big * 1024.0)
if wr:
    w = sorted(wr)[len(wr) // 2]
    out["calibration"]["write_probe_bytes"] = 131072000
    out["calibration"]["write_probe_WRITE_SIZE_KB"] = w
    out["calibration"]["write_factor"] = 131072000 / (w * 1024.0)
KERNEL = "sattn_fused_kernel"  # LN + q|k|v + self-attention of the 1000-token level in one launch (round 4; tools/one_op.py sattn_fused)
fa = per_dispatch("FETCH_SIZE_attn", "FETCH_SIZE", KERNEL)
wa = per_dispatch("WRITE_SIZE_attn", "WRITE_SIZE", KERNEL)
if fa and wa and rd and wr:
    f = sorted(fa)[len(fa) // 2] * 1024.0 * out["calibration"]["read_factor"]
    w = sorted(wa)[len(wa) // 2] * 1024.0 * out["calibration"]["write_factor"]
    out["kernel"] = KERNEL + "<bf16,D=32> LN + q|k|v + self-attention B'=64 heads=8 N=L=1000 (isolated probe)"
    out["raw_FETCH_SIZE_KB"] = sorted(fa)[len(fa) // 2]
    out["raw_WRITE_SIZE_KB"] = sorted(wa)[len(wa) // 2]
    out["read_bytes_per_launch"] = f
    out["write_bytes_per_launch"] = w
    out["traffic_bytes_per_launch"] = f + w
    out["algorithmic_bytes_per_launch"] = 2 * 64 * 1000 * 256 * 2 + 3 * 256 * 256 * 2  # x in, O out, the packed weights once


def totals(sub, counter):
    """(sum over every dispatch, {kernel: (dispatches, sum)}) of one pass"""
    dbs = glob.glob(os.path.join(root, sub, "**", "*.db"), recursive=True)
    if not dbs:
        return None, {}
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select kernel_name, count(distinct dispatch_id), sum(value) from counters_collection where counter_name = ? group by kernel_name",
                     (counter,)).fetchall()
    return sum(r[2] for r in rows), {r[0]: (r[1], r[2]) for r in rows}


# the whole captured step: bench.py --step-only with 1 and with 5 timed steps; (5 - 1) / 4 = one step (set-up, warm-up and capture cancel)
f1, fk1 = totals("FETCH_SIZE_step1", "FETCH_SIZE")
f5, fk5 = totals("FETCH_SIZE_step5", "FETCH_SIZE")
w1, wk1 = totals("WRITE_SIZE_step1", "WRITE_SIZE")
w5, wk5 = totals("WRITE_SIZE_step5", "WRITE_SIZE")
if None not in (f1, f5, w1, w5) and rd and wr:
    rf, wf = out["calibration"]["read_factor"], out["calibration"]["write_factor"]
    rbytes, wbytes = (f5 - f1) / 4.0 * 1024.0 * rf, (w5 - w1) / 4.0 * 1024.0 * wf
    out["whole_step"] = {"read_bytes_per_step": rbytes, "write_bytes_per_step": wbytes, "bytes_per_step": rbytes + wbytes,
                         "how": "bench.py --step-only (batch 32, La 32): (5-step pass - 1-step pass) / 4, each counter in its own pass"}
    import re
    per = {}
    for name in fk5:
        if name in fk1 and name in wk5 and name in wk1 and fk5[name][0] > fk1[name][0]:
            n = fk5[name][0] - fk1[name][0]
            per[name] = (n / 4.0, (fk5[name][1] - fk1[name][1]) / n * 1024.0 * rf, (wk5[name][1] - wk1[name][1]) / n * 1024.0 * wf)
    top = sorted(per.items(), key=lambda kv: -kv[1][0] * (kv[1][1] + kv[1][2]))[:24]
    out["per_kernel"] = [{"kernel": re.sub(r"\(anonymous namespace\)::", "", k)[:100], "launches_per_step": round(v[0], 1),
                          "read_bytes_per_launch": round(v[1]), "write_bytes_per_launch": round(v[2]),
                          "share_of_step_bytes": round(v[0] * (v[1] + v[2]) / (rbytes + wbytes), 4)} for k, v in top]
print(json.dumps(out, indent=1))
