# rocprofv3 kernel stats of the training step only: usage bash tools/train_profile.sh <tag> "<description>"
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_t}
DESC=${2:-"round 3"}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/proft_$TAG
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/proft_$TAG -o p -- python $R/bench.py --train --steps 5 --warmup 2 > $R/gpurun_out/proft_$TAG.log 2>&1
cd $R
DB=$(find gpurun_out/proft_$TAG -name "*.db" | head -1)
python tools/prof_summary.py $DB 8 "rocprofv3 --kernel-trace --stats -- python bench.py --train --steps 5 --warmup 2 ($DESC; cfg 5: per-GPU batch 4, bf16, graph-replayed micro-step)" > gpurun_out/${TAG}_train_kernel_stats.txt
head -30 gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-160
rm -rf gpurun_out/proft_$TAG
