#!/bin/bash
# the round's verification + evidence in one gpurun call: full GPU suite, kernel stats (step + train), bench line, PMC step, PMC traffic
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=${1:-r05_v3}
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 ) > gpurun_out/${TAG}_pytest.txt 2>&1
bash tools/round_profile.sh $TAG "round 5: geglu3 at C = 384, GELU in 13.5 instructions, packed feed-forward, in-place attn2" > gpurun_out/${TAG}_profile.log 2>&1
bash tools/round_pmc_step.sh $TAG "round 5" > gpurun_out/${TAG}_pmcstep.log 2>&1
timeout 1200 bash tools/round_pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmctraffic.log 2>&1
cat gpurun_out/${TAG}_pytest.txt; tail -c 600 gpurun_out/${TAG}_bench.json; head -12 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-140
