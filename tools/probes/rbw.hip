// Read-pattern probe: 64000 x 256 bf16 (32 MB).  (A) contiguous 16 B/lane; (B) MFMA-fragment pattern used by the
// row-panel kernels: lane (row = l&31, half = l>>5) loads 16 B at x[row][c*16 + half*8] for c = 0..15 (32 rows x 32 B
// per instruction); (C) coalesced rows: 64 lanes x 16 B = 2 rows x 512 B per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr long ROWS = 64000; constexpr int ROWB = 512;
__global__ void kA(const u32x4* x, unsigned* o, long n) { long i = (long)blockIdx.x * 256 + threadIdx.x; unsigned a = 0;
  if (i < n) { u32x4 v = x[i]; a = v[0] ^ v[1] ^ v[2] ^ v[3]; } if (a == 0x12345678u) o[0] = a; }
__global__ void kB(const unsigned char* x, unsigned* o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6; const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
  const unsigned char* p = x + (r0 + (lane & 31)) * ROWB + (lane >> 5) * 16; unsigned a = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) { u32x4 v = *(const u32x4*)(p + c * 32); a ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (a == 0x12345678u) o[0] = a; }
__global__ void kC(const unsigned char* x, unsigned* o) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6; const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
  const unsigned char* p = x + r0 * ROWB + lane * 16; unsigned a = 0;
#pragma unroll
  for (int c = 0; c < 16; ++c) { u32x4 v = *(const u32x4*)(p + c * 1024); a ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (a == 0x12345678u) o[0] = a; }
template <class F> float timeit(F f) { hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms / 20; }
int main() { unsigned char* x; unsigned* o; const long bytes = ROWS * ROWB; hipMalloc(&x, bytes * 8); hipMalloc(&o, 64); hipMemset(x, 1, bytes * 8);
  auto rep = [&](const char* n, float ms, long b) { printf("%-40s %8.1f us  %7.1f GB/s\n", n, ms * 1e3, b / ms / 1e6); };
  rep("A contiguous 32MB", timeit([&] { kA<<<dim3((unsigned)(bytes / 16 / 256)), 256>>>((u32x4*)x, o, bytes / 16); }), bytes);
  rep("B fragment pattern 32MB", timeit([&] { kB<<<dim3(ROWS / 128), 256>>>(x, o); }), bytes);
  rep("C coalesced rows 32MB", timeit([&] { kC<<<dim3(ROWS / 128), 256>>>(x, o); }), bytes);
  rep("A contiguous 256MB", timeit([&] { kA<<<dim3((unsigned)(bytes * 8 / 16 / 256)), 256>>>((u32x4*)x, o, bytes * 8 / 16); }), bytes * 8);
  return 0; }
