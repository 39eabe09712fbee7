cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
APAD_LIB_PATH=$PWD/exp/lib_g3reg.so timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "geglu_packed" 2>&1 | tail -3
APAD_LIB_PATH=$PWD/exp/lib_g3reg.so timeout 300 python tools/g3_stress.py 1 2>&1 | tail -2
timeout 400 bash tools/kt.sh geglu3 geglu3_kernel product g3reg product g3reg
APAD_LIB_PATH=exp/lib_g3regtr.so timeout 200 python tools/g3_trace.py
bash tools/ab_step.sh "APAD_LIB_PATH=exp/lib_g3reg.so" "APAD_X=1"
} > gpurun_out/r5_geglu3.log 2>&1
tail -50 gpurun_out/r5_geglu3.log
