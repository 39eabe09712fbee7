# same-box A/B: low-res section on one stream / two streams x tiled / ring GEMM tiles.  ARMS="streams ring,streams ring,..."
cd ${GRAFT_REPO_ROOT:-/root/repo}
IFS="," read -ra LIST <<< "${ARMS:-0 0,0 1,1 0,1 1}"
for rep in 1 2; do for arm in "${LIST[@]}"; do
  set -- $arm
  ms=$(APAD_GEMM_RING=$2 python bench.py --step-only --steps 30 --warmup 3 --low-res-streams $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "low-res-streams $1 ring $2 -> $ms ms"; done; done
