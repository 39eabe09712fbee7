#!/bin/bash
# the round's evidence in one gpurun call: kernel stats (step + train), bench line, PMC step, PMC traffic, PMC of the north-star kernel and of the
# halo convolution, the operand-data MFMA microbenchmark.  usage: bash tools/r6_final.sh <tag>   (the GPU suite runs in its own call)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
TAG=${1:-r06_v1}
exp/mfma_data > gpurun_out/${TAG}_ubench_mfma_data.txt 2>&1
bash tools/round_profile.sh $TAG "round 6: halo-resident 3x3 convolutions (csrc/hconv.hip), K-sliced 64-token-level convolutions, GroupNorm statistics finalised once, attn2 with <= 512 audio keys on the fused routes of every level, sattn_fused: two workgroups per CU at d = 48, statistics on the dot-product instruction, LDS-staged token fragments" > gpurun_out/${TAG}_profile.log 2>&1
bash tools/round_pmc_step.sh $TAG "round 6" > gpurun_out/${TAG}_pmcstep.log 2>&1
timeout 1200 bash tools/round_pmc_traffic.sh $TAG > gpurun_out/${TAG}_pmctraffic.log 2>&1
for spec in "xattn xattn_kernel" "hconv256 hconv_kernel" "hconv128 hconv_kernel"; do set -- $spec
  timeout 600 bash tools/pmc_kernel.sh $1 $2 ${TAG}_$1 > gpurun_out/${TAG}_pmc_$1.txt 2>&1
  rm -rf gpurun_out/pmc_${TAG}_$1; done   # (the counter databases: tens of MB each; gpurun copies back at most 64 MiB)
find gpurun_out -name "*.db" -delete 2>/dev/null; du -sh gpurun_out
tail -c 600 gpurun_out/${TAG}_bench.json; head -12 gpurun_out/${TAG}_bench_kernel_stats.txt | cut -c1-140
