// MFMA-stream microbenchmark: one wave per SIMD, 48 MFMA 32x32x16 bf16 per iteration (two accumulator chains) fed by ds_read_b128 fragments.
//   RD: 0 no LDS reads (fragments stay in registers), 1 one step ahead, 2 two steps ahead, 3 four reads ahead in pairs (geglu3's m3_read2 pattern)
//   build: hipcc --offload-arch=gfx950 -O3 -o exp/mstream tools/ubench/mstream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float V16 __attribute__((ext_vector_type(16)));
typedef short B8 __attribute__((ext_vector_type(8)));
typedef uint32_t U4 __attribute__((ext_vector_type(4)));

template <int RD, bool BA, int WAVES>
__global__ __launch_bounds__(512) void k(uint64_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[65536];
    const int lane = threadIdx.x & 63;
    uint32_t laddr = (uint32_t)(uintptr_t)sm + lane * 16;
    uint64_t t0 = 0, t1 = 0;
    V16 acc0 = {0}, acc1 = {0};
    B8 b0, b1;
    for (int i = 0; i < 8; ++i) { b0[i] = (short)(threadIdx.x + i); b1[i] = (short)(threadIdx.x * 3 + i); }
    if (BA) asm volatile("" : "+a"(b0), "+a"(b1));
    U4 f[4];
    for (int i = 0; i < 4; ++i) f[i] = U4{(uint32_t)lane, 1, 2, 3};
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#define MF(acc, a, b) do { if (BA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b)); else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); } while (0)
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (RD == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(laddr));
        if (RD == 2) { asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(laddr)); asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(f[1]) : "v"(laddr)); }
        if (RD == 3) { asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(laddr)); asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(f[1]) : "v"(laddr)); }
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            if (RD == 1) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(s + 1) & 1]) : "v"(laddr), "n"(1024));
                asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            }
            if (RD == 2) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(s + 2) & 3]) : "v"(laddr), "n"(2048));
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            }
            if (RD == 3 && (s & 1) == 0) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(s + 2) & 3]) : "v"(laddr), "n"(2048));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(s + 3) & 3]) : "v"(laddr), "n"(3072));
                asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            }
            B8 a = __builtin_bit_cast(B8, f[RD == 1 ? (s & 1) : (s & 3)]);
            MF(acc0, a, b0);
            MF(acc1, a, b1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n s_nop 7\n s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sink = 0;
    for (int i = 0; i < 16; ++i) sink += acc0[i] + acc1[i];
    if (sink == 1234.5678f) out[1000000] = 1;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int RD, bool BA, int WAVES>
void run(uint64_t* dout, int iters = 500) {
    const int nb = 256;
    k<RD, BA, WAVES><<<nb, 64 * WAVES>>>(dout, iters);
    k<RD, BA, WAVES><<<nb, 64 * WAVES>>>(dout, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nb);
    hipMemcpy(h.data(), dout, nb * 8, hipMemcpyDeviceToHost);
    double a = 0;
    for (int i = 0; i < nb; ++i) a += (double)h[i];
    printf("RD=%d B in %s waves/CU=%d: %7.0f cycles per 48 MFMA (%.1f per MFMA)\n", RD, BA ? "AGPR" : "VGPR", WAVES, a / nb / iters, a / nb / iters / 48);
}

int main() {
    uint64_t* dout;
    hipMalloc(&dout, 8 * 1000016);
    run<0, false, 4>(dout); run<0, true, 4>(dout);
    run<1, false, 4>(dout); run<1, true, 4>(dout);
    run<2, false, 4>(dout); run<2, true, 4>(dout);
    run<3, false, 4>(dout); run<3, true, 4>(dout);
    run<0, true, 1>(dout); run<1, true, 1>(dout); run<2, true, 1>(dout); run<3, true, 1>(dout);
    run<1, true, 8>(dout); run<3, true, 8>(dout);
    return 0;
}
