#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
bash tools/round_profile.sh r05_v2 "round 5: packed feed-forward, in-place attn2, one stream at the 64-token level" > gpurun_out/r05_v2_profile.log 2>&1; tail -3 gpurun_out/r05_v2_profile.log
bash tools/round_pmc_step.sh r05_v2 "round 5" > gpurun_out/r05_v2_pmcstep.log 2>&1; tail -3 gpurun_out/r05_v2_pmcstep.log
bash tools/round_pmc_traffic.sh r05_v2 > gpurun_out/r05_v2_traffic.log 2>&1; tail -5 gpurun_out/r05_v2_traffic.log
