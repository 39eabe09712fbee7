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
KERNEL = "attn2q_kernel"  # the self-attention kernel of the 1000-token level (two query tiles per wave since round 2)
fa = per_dispatch("FETCH_SIZE_attn", "FETCH_SIZE", KERNEL)
wa = per_dispatch("WRITE_SIZE_attn", "WRITE_SIZE", KERNEL)
if fa and wa and rd and wr:
    f = sorted(fa)[len(fa) // 2] * 1024.0 * out["calibration"]["read_factor"]
    w = sorted(wa)[len(wa) // 2] * 1024.0 * out["calibration"]["write_factor"]
    out["kernel"] = KERNEL + "<bf16,D=32> self-attention B'=64 heads=8 N=L=1000"
    out["raw_FETCH_SIZE_KB"] = sorted(fa)[len(fa) // 2]
    out["raw_WRITE_SIZE_KB"] = sorted(wa)[len(wa) // 2]
    out["read_bytes_per_launch"] = f
    out["write_bytes_per_launch"] = w
    out["traffic_bytes_per_launch"] = f + w
    out["algorithmic_bytes_per_launch"] = 131072000
print(json.dumps(out, indent=1))
