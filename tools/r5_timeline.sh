R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_tl -o p -- python $R/bench.py --steps 5 --warmup 1 --step-only > /tmp/prof_tl.log 2>&1
DB=$(find /tmp/prof_tl -name "*.db" | head -1)
cd $R; python tools/step_timeline.py $DB > gpurun_out/r5_timeline.txt 2>&1
head -60 gpurun_out/r5_timeline.txt | cut -c1-170
