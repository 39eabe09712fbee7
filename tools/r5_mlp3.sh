cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "geglu_mlp" 2>&1 | tail -3
timeout 400 bash tools/kt.sh mlp3 mlp3_kernel old product old product
bash tools/ab_step.sh "APAD_LIB_PATH=exp/lib_old.so" "APAD_X=1"
} > gpurun_out/r5_mlp3.log 2>&1
tail -30 gpurun_out/r5_mlp3.log
