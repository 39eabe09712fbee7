# same-box A/B of the captured step: usage bash tools/ab_step.sh "ENV1=a ENV2=b" "ENV1=c" ...   (each argument = one arm's environment)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for arm in "$@"; do
  ms=$(env $arm python bench.py --step-only --steps 30 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "$arm -> $ms ms"; done; done
