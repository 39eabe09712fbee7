#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run3; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "fused_cross_attention" > $O/t_xattn.log 2>&1; tail -5 $O/t_xattn.log
timeout 900 python -m pytest tests/test_gpu_processors.py -x -q -k "block_entry" > $O/t_proc.log 2>&1; tail -5 $O/t_proc.log
timeout 600 python tools/attn2_grid.py > $O/grid.json 2>$O/grid.err; cat $O/grid.json | tr -d '\n' | sed 's/},/},\n/g'
