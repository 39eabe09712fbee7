cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-/root/repo}
( cd /tmp && export TMPDIR=/tmp && for op in geglu3 rp_geglu384; do rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$op -o p -- python $R/tools/one_op.py $op 30 > /dev/null 2>&1; f=$(find $R/gpurun_out/kt_$op -name "*kernel_stats.csv" | head -1); head -4 $f; done ) > gpurun_out/r5_geglu3b.log 2>&1
bash tools/pmc_kernel.sh geglu3 geglu3_kernel >> gpurun_out/r5_geglu3b.log 2>&1
cat gpurun_out/r5_geglu3b.log
