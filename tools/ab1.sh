R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for cfg in "2 1" "4 1" "8 1" "2 2" "4 2" "1 0"; do set -- $cfg
  echo "== NSTREAMS=$1 levels=$2"; APAD_LOW_RES_NSTREAMS=$1 python bench.py --step-only --steps 30 --warmup 3 --low-res-streams $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done
python tools/gemm_small_bench.py 2>&1 | tail -20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/tl -o p -- python $R/bench.py --steps 3 --warmup 1 --step-only > $R/gpurun_out/tl.log 2>&1
cd $R; DB=$(find gpurun_out/tl -name "*.db" | head -1); python tools/step_timeline.py $DB --seq > gpurun_out/timeline.txt; head -50 gpurun_out/timeline.txt; rm -rf gpurun_out/tl
