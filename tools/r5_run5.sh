#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run5; mkdir -p $O
timeout 600 python tools/np_modes.py 4 > $O/np_modes.log 2>&1; tail -8 $O/np_modes.log
( time timeout 1500 python -m pytest tests -q -m gpu --durations=15 ) > $O/suite.log 2>&1; tail -30 $O/suite.log
timeout 300 python tools/attn2_grid.py 2>/dev/null | tr -d '\n' | sed 's/},/},\n/g' | grep C256
