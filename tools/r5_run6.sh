#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "geglu_mlp or fused_cross" > $O/t1.log 2>&1; tail -8 $O/t1.log
timeout 1200 python -m pytest tests/test_gpu_unet.py -q -k "independent_of_their_batch or cfg_shared_prefix" > $O/t2.log 2>&1; tail -12 $O/t2.log
timeout 900 python -m pytest tests/test_gpu_processors.py -q > $O/t3.log 2>&1; tail -3 $O/t3.log
timeout 600 python tools/np_modes.py 32 > $O/np_modes.log 2>&1; tail -8 $O/np_modes.log
timeout 300 python tools/mlp_bench.py 64000 32000 16000 2>/dev/null | grep "M="
