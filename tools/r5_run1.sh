#!/bin/bash
# round 5, first GPU call: the packed feed-forward kernel (parity + timing + ablations), the round-5 goldens, a step-level A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "geglu_mlp" > $O/t_mlp.log 2>&1; tail -3 $O/t_mlp.log
timeout 300 python tools/mlp_bench.py 64000 32000 48000 > $O/mlp_bench.log 2>&1
for a in 1 6 7 8 16; do APAD_LIB_PATH=exp/lib_m3abl$a.so timeout 120 python tools/mlp_bench.py 64000 >> $O/mlp_bench.log 2>&1; done
cat $O/mlp_bench.log
timeout 900 python -m pytest tests/test_gpu_processors.py -x -q > $O/t_proc.log 2>&1; tail -3 $O/t_proc.log
bash tools/ab_step.sh "APAD_MLP_PACKED=0" "APAD_MLP_PACKED=1" > $O/ab_mlp.log 2>&1; cat $O/ab_mlp.log
