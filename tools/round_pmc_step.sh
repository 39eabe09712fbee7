R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_step_r02
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $R/gpurun_out/pmc_step_r02 -o p -- python $R/bench.py --steps 1 --warmup 1 --step-only > $R/gpurun_out/pmc_step_r02.log 2>&1
cd $R
DB=$(find gpurun_out/pmc_step_r02 -name "*.db" | head -1)
echo "# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python bench.py --steps 1 --warmup 1 --step-only (round 2: attn2q_kernel, mlp2_kernel, 8-wave row-panel workgroups; per-kernel sums; percentages are of the kernel's own wave-cycles)" > gpurun_out/r02_pmc_step_v1.txt
python tools/pmc_step_summary.py $DB >> gpurun_out/r02_pmc_step_v1.txt
head -16 gpurun_out/r02_pmc_step_v1.txt | cut -c1-170
rm -rf gpurun_out/pmc_step_r02
