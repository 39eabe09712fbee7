#!/bin/bash
# kernel-trace timing of ONE op (tools/one_op.py) under a variant library: usage bash tools/kt.sh <op> <substring> [lib tag ...]   ("product" = the in-tree library)
R=${GRAFT_REPO_ROOT:-/root/repo}; OP=$1; SUB=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  rm -rf /tmp/kt_$tag
  if [ "$tag" = product ]; then unset APAD_LIB_PATH; else export APAD_LIB_PATH=$R/exp/lib_$tag.so; fi
  timeout 120 rocprofv3 --kernel-trace -d /tmp/kt_$tag -o p -- python $R/tools/one_op.py $OP 20 > /tmp/kt_$tag.log 2>&1
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  echo "== $tag"; [ -n "$db" ] && python $R/tools/kt_avg.py $db "$SUB" || tail -3 /tmp/kt_$tag.log
done
