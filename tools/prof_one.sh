# rocprofv3 kernel stats of the captured step under an environment: usage bash tools/prof_one.sh <tag> "ENV=..." [grep pattern]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; ENVS=$2; PAT=${3:-.}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
env $ENVS rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o p -- python $R/bench.py --steps 5 --warmup 1 --step-only > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
DB=$(find gpurun_out/prof_$TAG -name "*.db" | head -1)
python tools/prof_summary.py $DB 7 "rocprofv3 --kernel-trace --stats ($ENVS)" > gpurun_out/${TAG}_kernel_stats.txt
grep -E "$PAT" gpurun_out/${TAG}_kernel_stats.txt | head -20 | cut -c1-150
rm -rf gpurun_out/prof_$TAG
