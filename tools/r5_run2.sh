#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out; O=gpurun_out/r5_run2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "geglu_mlp" > $O/t_mlp.log 2>&1; tail -3 $O/t_mlp.log
timeout 300 python tools/mlp_bench.py 64000 > $O/mlp_bench.log 2>&1
for a in m3old m3imm m3bare m3noz m3abl8; do APAD_LIB_PATH=exp/lib_$a.so timeout 120 python tools/mlp_bench.py 64000 >> $O/mlp_bench.log 2>&1; done
grep "M=" $O/mlp_bench.log
timeout 900 bash tools/pmc_kernel.sh mlp3 mlp3_kernel r5_mlp3 > $O/pmc.log 2>&1; cat gpurun_out/pmc_r5_mlp3/summary.txt
