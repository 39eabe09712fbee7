R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_r02b
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r02b -o p -- python $R/bench.py --steps 5 --warmup 1 --step-only > $R/gpurun_out/prof_r02b.log 2>&1
cd $R
DB=$(find gpurun_out/prof_r02b -name "*.db" | head -1)
python tools/prof_summary.py $DB 7 "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --step-only (round 2, v3 = HEAD: two-query-tile self-attention, two-wave feed-forward kernel, 8-wave row-panel workgroups, MI355X, batch 32, La=32)" > gpurun_out/r02_bench_kernel_stats_v3.txt
rm -rf gpurun_out/traffic
bash tools/pmc_traffic.sh > gpurun_out/r02_pmc_traffic.json 2> gpurun_out/r02_pmc_traffic.err
python bench.py > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err
tail -c 600 gpurun_out/bench_r02b.json; head -12 gpurun_out/r02_bench_kernel_stats_v3.txt | cut -c1-150; cat gpurun_out/r02_pmc_traffic.json | head -30
# keep the scratch directory small
rm -rf gpurun_out/prof_r02b gpurun_out/traffic
