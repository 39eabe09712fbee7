# timing-only A/B of csrc/cgemm.hip builds (tools/ab_build.sh): usage bash tools/cgemm_ab.sh <tag> ...   (tag "bn128": the product build with APAD_CGEMM_BN=128)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== product"; CGEMM_CHILD=1 CGEMM_NOCHECK=1 APAD_CGEMM=1 python tools/cgemm_bench.py 2>&1 | grep -E "^conv|^gemm" | cut -c1-75
for t in "$@"; do echo "== $t"
  if [ $t = bn128 ]; then CGEMM_CHILD=1 CGEMM_NOCHECK=1 APAD_CGEMM=1 APAD_CGEMM_BN=128 python tools/cgemm_bench.py 2>&1 | grep -E "^conv|^gemm" | cut -c1-75
  else CGEMM_CHILD=1 CGEMM_NOCHECK=1 APAD_CGEMM=1 APAD_LIB_PATH=exp/lib_$t.so python tools/cgemm_bench.py 2>&1 | grep -E "^conv|^gemm" | cut -c1-75; fi; done
