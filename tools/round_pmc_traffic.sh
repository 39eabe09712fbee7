#!/bin/bash
# HBM traffic of the WHOLE captured step and of its kernels from the PMC counters (run on the GPU box through gpurun; counters only, no
# trace domains): FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each over bench.py --step-only with 1 and with 5 timed steps
# (their difference / 4 = bytes of one step, set-up and warm-up cancel), over the self-attention probe and over a read / write probe of KNOWN
# byte count (the gfx950 unit corrections of MI355X_MICROARCH.md's HBM section, calibrated in the same run).
#   usage: bash tools/round_pmc_traffic.sh <tag e.g. r04_v1>   ->   gpurun_out/<tag>_pmc_traffic.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r04_v1}
OUT=$R/gpurun_out/traffic_$TAG
rm -rf $OUT; mkdir -p $OUT
[ -x $R/tools/probes/rbw ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $R/tools/probes/rbw $R/tools/probes/rbw.hip
[ -x $R/tools/probes/wbw ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $R/tools/probes/wbw $R/tools/probes/wbw.hip
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for n in 1 5; do
    for try in 1 2 3; do  # (a counter pass over the graph-replayed step occasionally never returns on this ROCm: bounded, retried)
      rm -rf $OUT/${c}_step$n
      APAD_LOW_RES_NSTREAMS=1 timeout 150 rocprofv3 --pmc $c -d $OUT/${c}_step$n -o p -- python $R/bench.py --steps $n --warmup 1 --step-only > $OUT/${c}_step$n.log 2>&1 && break
      echo "pass ${c}_step$n try $try failed / timed out"
    done
  done
  timeout 150 rocprofv3 --pmc $c -d $OUT/${c}_attn -o p -- python $R/tools/one_op.py sattn_fused 5 > $OUT/${c}_attn.log 2>&1
  timeout 120 rocprofv3 --pmc $c -d $OUT/${c}_rbw -o p -- $R/tools/probes/rbw > $OUT/${c}_rbw.log 2>&1
  timeout 120 rocprofv3 --pmc $c -d $OUT/${c}_wbw -o p -- $R/tools/probes/wbw > $OUT/${c}_wbw.log 2>&1
done
cd $R && python tools/pmc_traffic_summary.py $OUT > gpurun_out/${TAG}_pmc_traffic.json
cat gpurun_out/${TAG}_pmc_traffic.json | head -80
rm -rf $OUT
