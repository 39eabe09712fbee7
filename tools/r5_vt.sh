cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_processors.py -q -x -m gpu -k "attention or attn or self or golden" 2>&1 | tail -5
for t in old product old product; do if [ $t = product ]; then unset APAD_LIB_PATH; else export APAD_LIB_PATH=$PWD/exp/lib_$t.so; fi; timeout 120 python tools/sattn_bench.py 2>/dev/null | grep sattn; done
unset APAD_LIB_PATH
bash tools/ab_step.sh "APAD_LIB_PATH=exp/lib_old.so" "APAD_X=1"
} > gpurun_out/r5_vt.log 2>&1
tail -30 gpurun_out/r5_vt.log
