#!/bin/bash
# HBM traffic of the roofline kernel from the PMC counters (run on the GPU box through gpurun):
#   FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each also over a read / write probe of KNOWN byte count
#   so the gfx950 unit corrections (MI355X_MICROARCH.md, HBM section) are calibrated in the same run.
# Result: gpurun_out/traffic/*.db -> tools/pmc_traffic_summary.py -> profiles/rNN_pmc_traffic.json
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/${c}_attn -o p -- python $R/tools/one_op.py attn_self 5 > $OUT/${c}_attn.log 2>&1
  rocprofv3 --pmc $c -d $OUT/${c}_rbw -o p -- $R/tools/probes/rbw > $OUT/${c}_rbw.log 2>&1
  rocprofv3 --pmc $c -d $OUT/${c}_wbw -o p -- $R/tools/probes/wbw > $OUT/${c}_wbw.log 2>&1
done
cd $R && python tools/pmc_traffic_summary.py $OUT
