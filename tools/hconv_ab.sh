# timing-only A/B of csrc/hconv.hip builds (tools/ab_build.sh): usage bash tools/hconv_ab.sh <tag> ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== product"; CGEMM_CHILD=1 CGEMM_NOCHECK=1 python tools/cgemm_bench.py 2>&1 | grep -E "^conv" | cut -c1-75
for t in "$@"; do echo "== $t"
  CGEMM_CHILD=1 CGEMM_NOCHECK=1 APAD_LIB_PATH=exp/lib_$t.so python tools/cgemm_bench.py 2>&1 | grep -E "^conv" | cut -c1-75; done
