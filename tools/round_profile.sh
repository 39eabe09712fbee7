# One gpurun call: rocprofv3 kernel stats of the captured step and of the training step, the default bench line.
#   usage: bash tools/round_profile.sh <tag, e.g. r03_v1> "<description>"
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_v1}
DESC=${2:-"round 3"}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/proft_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o p -- python $R/bench.py --steps 5 --warmup 1 --step-only > $R/gpurun_out/prof_$TAG.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/proft_$TAG -o p -- python $R/bench.py --train --steps 5 --warmup 2 > $R/gpurun_out/proft_$TAG.log 2>&1
cd $R
DB=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
python tools/prof_summary.py $DB 7 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --step-only ($DESC; MI355X, batch 32, La=32)" > gpurun_out/${TAG}_bench_kernel_stats.txt
DB=$(find gpurun_out/proft_$TAG -name "*.db" | head -1)
python tools/prof_summary.py $DB 8 "rocprofv3 --kernel-trace --stats -- python bench.py --train --steps 5 --warmup 2 ($DESC; cfg 5: per-GPU batch 4, bf16, graph-replayed micro-step)" > gpurun_out/${TAG}_train_kernel_stats.txt
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 1500 gpurun_out/${TAG}_bench.json; head -14 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-150; head -24 gpurun_out/${TAG}_train_kernel_stats.txt | cut -c1-150
rm -rf gpurun_out/prof_$TAG gpurun_out/proft_$TAG
