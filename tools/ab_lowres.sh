cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for cfg in "1" "0" "2"; do
  ms=$(python bench.py --step-only --steps 30 --warmup 3 --low-res-streams $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "low-res-streams $cfg -> $ms ms"; done; done
