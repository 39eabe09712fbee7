# usage: bash tools/round_pmc_step.sh <tag, e.g. r03_v1> "<description>"   (counters only: no --stats / trace domains beside --pmc)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_v1}
DESC=${2:-"round 3"}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_step_$TAG
CNT="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"
timeout 600 rocprofv3 --pmc $CNT -d $R/gpurun_out/pmc_step_$TAG -o p -- python $R/bench.py --steps 1 --warmup 1 --step-only > $R/gpurun_out/pmc_step_$TAG.log 2>&1
cd $R
DB=$(find gpurun_out/pmc_step_$TAG -name "*.db" | head -1)
echo "# rocprofv3 --pmc $CNT -- python bench.py --steps 1 --warmup 1 --step-only ($DESC; per-kernel sums; percentages are of the kernel's own wave-cycles)" > gpurun_out/${TAG}_pmc_step.txt
python tools/pmc_step_summary.py $DB >> gpurun_out/${TAG}_pmc_step.txt
head -16 gpurun_out/${TAG}_pmc_step.txt | cut -c1-170
rm -rf gpurun_out/pmc_step_$TAG
