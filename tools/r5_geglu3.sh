cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
bash tools/ab_step.sh "APAD_GEGLU3_MIN_M=12000" "APAD_GEGLU3_MIN_M=99999999"
timeout 1200 python -m pytest tests -q -x -m gpu -k "batch_independ or shared_prefix or golden or geglu" 2>&1 | tail -5
} > gpurun_out/r5_geglu3.log 2>&1
tail -50 gpurun_out/r5_geglu3.log
