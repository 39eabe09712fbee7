#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run7; mkdir -p $O
NP_STREAMS=2 timeout 600 python tools/np_modes.py 32 > $O/np_modes_streams.log 2>&1; tail -6 $O/np_modes_streams.log
timeout 900 python bench.py --no-cpu-baseline --no-train-leg > $O/bench_notrain.json 2>$O/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5_run7/bench_notrain.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("noise_pred_vs_fp32_mode"), d.get("f16_mode",{}).get("ms_per_step"), d.get("fp32_mode",{}).get("ms_per_step"))
PY
