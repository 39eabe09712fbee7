#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run4; mkdir -p $O
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=30 ) > $O/suite.log 2>&1; tail -45 $O/suite.log
bash tools/ab_step.sh "APAD_LOW_RES_NSTREAMS=2" "APAD_LOW_RES_NSTREAMS=1" > $O/ab_streams.log 2>&1; cat $O/ab_streams.log
bash tools/round_profile.sh r05_v1 "round 5: packed feed-forward kernel, chunked attn2" > $O/profile.log 2>&1; tail -60 $O/profile.log
