#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "fused_cross_attention or pack" > $O/t1.log 2>&1; tail -6 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_processors.py -q > $O/t2.log 2>&1; tail -4 $O/t2.log
timeout 900 python -m pytest tests/test_gpu_unet.py -q -k "one_launch or fused or graph or scale_zero" > $O/t3.log 2>&1; tail -4 $O/t3.log
timeout 300 python tools/attn2_grid.py 2>/dev/null | tr -d '\n' | sed 's/},/},\n/g' | grep C256
bash tools/ab_step.sh "APAD_FUSED_XATTN=1" > $O/step.log 2>&1; cat $O/step.log
