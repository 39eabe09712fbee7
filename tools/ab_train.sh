# same-box A/B of the training step (cfg 5): usage bash tools/ab_train.sh "ENV=a" "ENV=b" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for arm in "$@"; do
  ms=$(env $arm python bench.py --train --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "$arm -> $ms ms"; done; done
