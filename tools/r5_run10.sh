#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "fused_cross_attention" 2>&1 | tail -2
for l in product m3pk mlpv2all; do if [ $l = product ]; then timeout 200 python tools/mlp_bench.py 64000 16000 8000 4000 1000; else APAD_LIB_PATH=exp/lib_$l.so timeout 200 python tools/mlp_bench.py 64000 16000 8000 4000 1000; fi; done 2>/dev/null | grep "M=" | tee $O/mlp.log
timeout 300 python tools/attn2_grid.py 2>/dev/null | tr -d '\n' | sed 's/},/},\n/g' | grep C256
