cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -q -x -m gpu -k "geglu or mlp or gelu or batch_independ or shared_prefix or golden or kernel_form or hs_ or feed" 2>&1 | tail -8
timeout 300 bash tools/kt.sh mlp3 mlp3_kernel product
timeout 300 bash tools/kt.sh geglu3 geglu3_kernel product
python bench.py --step-only --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step', d['ms_per_step'])"
} > gpurun_out/r5_gelu.log 2>&1
tail -30 gpurun_out/r5_gelu.log
