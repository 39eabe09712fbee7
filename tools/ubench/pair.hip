// Two waves per SIMD, each running geglu3's per-chunk pattern: 24 MFMA 32x32x16 bf16 into ONE accumulator (a dependent chain) with NV vector
// instructions behind each of the first 16; optional workgroup barrier per iteration; the slowest wave's span per iteration.
//   CH: accumulator chains per wave (1 = every MFMA depends on the one before);  build: hipcc --offload-arch=gfx950 -O3 -o exp/pair tools/ubench/pair.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef float V16 __attribute__((ext_vector_type(16)));
typedef short B8 __attribute__((ext_vector_type(8)));

template <int NV, int NT, int CH, bool BAR, bool SHIFT>
__global__ __launch_bounds__(512) void k(uint64_t* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    V16 acc[2] = {{0}, {0}};
    B8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    asm volatile("" : "+a"(b));
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    float c1 = 1.0001f, c2 = 0.5f;
    uint64_t t0, t1;
    const int first = SHIFT && wave >= 4 ? 8 : 0;  // wave-uniform: MFMA slots [first, first + 16) carry the vector work
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s % CH]) : "v"(a), "a"(b));
            if (first == 0 ? s < 16 : s >= 8) {
#pragma unroll
                for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(c1), "v"(c2));
#pragma unroll
                for (int j = 0; j < NT; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(j + 4) & 7]));
            }
        }
    }
    asm volatile("s_nop 7\n s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sink = 0;
    for (int i = 0; i < 16; ++i) sink += acc[0][i] + acc[1][i];
    for (int i = 0; i < 8; ++i) sink += x[i];
    if (sink == 1234.5678f) out[1000000] = 1;
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

template <int NV, int NT, int CH, bool BAR, bool SHIFT>
void run(uint64_t* dout, int waves) {
    const int nb = 256, iters = 400;
    k<NV, NT, CH, BAR, SHIFT><<<nb, waves * 64>>>(dout, iters);
    k<NV, NT, CH, BAR, SHIFT><<<nb, waves * 64>>>(dout, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nb * 16);
    hipMemcpy(h.data(), dout, nb * 16 * 8, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int bb = 0; bb < nb; ++bb) {
        uint64_t lo = ~0ull, hi = 0;
        for (int w = 0; w < waves; ++w) { lo = std::min(lo, h[(bb * 8 + w) * 2]); hi = std::max(hi, h[(bb * 8 + w) * 2 + 1]); }
        tot += (double)(hi - lo);
    }
    printf("waves/SIMD=%d NV=%d NT=%d chains=%d bar=%d shift=%d: %7.0f ticks per iteration (24 MFMA per wave; MFMA pipe alone = %d)\n", waves / 4, NV, NT, CH, (int)BAR,
           (int)SHIFT, tot / nb / iters, waves / 4 * 24 * 32);
}

int main() {
    uint64_t* dout;
    hipMalloc(&dout, 8 * 1000016);
    run<0, 0, 1, false, false>(dout, 4); run<0, 0, 1, false, false>(dout, 8); run<0, 0, 2, false, false>(dout, 4); run<0, 0, 2, false, false>(dout, 8);
    run<7, 1, 1, false, false>(dout, 4); run<7, 1, 1, false, false>(dout, 8); run<7, 1, 1, true, false>(dout, 8); run<7, 1, 1, true, true>(dout, 8);
    run<7, 1, 2, false, false>(dout, 8); run<7, 1, 2, true, false>(dout, 8); run<7, 1, 2, true, true>(dout, 8);
    run<4, 0, 1, false, false>(dout, 8); run<4, 0, 1, true, false>(dout, 8);
    return 0;
}
