// LDS read bandwidth per CU: every wave issues ds_read_b128 (1 KB per instruction, conflict-free) back to back; the slowest wave's s_memtime span.
//   build: hipcc --offload-arch=gfx950 -O3 -o exp/ldsbw tools/ubench/ldsbw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
typedef uint32_t U2 __attribute__((ext_vector_type(2)));
template <int W>
__global__ __launch_bounds__(1024) void k(uint64_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t laddr = (uint32_t)(uintptr_t)sm + lane * W + wave * 2048;
    U4 d[8];
    for (int i = 0; i < 8; ++i) d[i] = U4{0, 0, 0, 0};
    __syncthreads();
    uint64_t t0, t1;
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (W == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[j]) : "v"(laddr), "n"(1024));
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*(U2*)&d[j]) : "v"(laddr), "n"(1024));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += d[i][0];
    if (s == 0x12345678u) out[1000000] = 1;
    if (lane == 0) { out[(blockIdx.x * 16 + wave) * 2] = t0; out[(blockIdx.x * 16 + wave) * 2 + 1] = t1; }
}
template <int W> void run(uint64_t* dout, int waves) {
    const int nb = 256, iters = 2000;
    k<W><<<nb, waves * 64>>>(dout, iters);
    k<W><<<nb, waves * 64>>>(dout, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nb * 32);
    hipMemcpy(h.data(), dout, nb * 32 * 8, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int b = 0; b < nb; ++b) {
        uint64_t lo = ~0ull, hi = 0;
        for (int w = 0; w < waves; ++w) { lo = std::min(lo, h[(b * 16 + w) * 2]); hi = std::max(hi, h[(b * 16 + w) * 2 + 1]); }
        tot += (double)(hi - lo);
    }
    const double cyc = tot / nb, bytes = (double)waves * iters * 8 * 64 * W;
    printf("ds_read_b%d, %2d waves/CU: %.1f bytes / cycle / CU\n", W * 8, waves, bytes / cyc);
}
int main() {
    uint64_t* dout;
    hipMalloc(&dout, 8 * 1000016);
    for (int w : {1, 2, 4, 8, 16}) run<16>(dout, w);
    for (int w : {4, 8, 16}) run<8>(dout, w);
    return 0;
}
