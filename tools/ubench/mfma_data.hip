// What the dense bf16 MFMA rate depends on besides the instruction stream: the OPERAND DATA.  Four independent v_mfma_f32_32x32x16_bf16 accumulator
// chains per wave, two waves per SIMD, no memory traffic; the operand registers hold (a) zeros, (b) one constant, (c) eight rotating register sets of
// random bf16 values in [-1, 1).  Prints TF/s and the shader clock the wave saw (s_memtime ticks per microsecond of the 100 MHz wall clock), for a short
// (~0.2 ms) and a long (~20 ms) launch: on MI355X the matrix pipe is power-managed -- random operands draw more, the clock drops, and that clock (not 2.4 GHz)
// is what every MFMA-bound kernel of the step runs at.
//   build: hipcc --offload-arch=gfx950 -O3 -o exp/mfma_data tools/ubench/mfma_data.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int MODE> __global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
    bf16x8 a[8], b[8];
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 8; ++i) {
            float va = 0.f, vb = 0.f;
            if (MODE == 1) { va = 0.37f; vb = -0.61f; }
            if (MODE == 2) {
                va = (float)(hash32(threadIdx.x * 131u + s * 17u + i) & 0xffff) / 32768.f - 1.f;
                vb = (float)(hash32(threadIdx.x * 977u + s * 29u + i + 7u) & 0xffff) / 32768.f - 1.f;
            }
            a[s][i] = (__bf16)va;
            b[s][i] = (__bf16)vb;
        }
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[(s + c) & 7], acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE> void run(const char* what, int iters, float* d, unsigned long long* clk) {
    const int blocks = 256 * 2;  // two 4-wave workgroups per CU = two waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, clk, 50);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 16;
    printf("%-28s %7.3f ms  %7.1f TF/s  = %.3f of 2500   shader clock %.0f MHz (s_memtime ticks per 100 MHz wall tick x 100)\n", what, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 2500.0, h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0);
}

int main() {
    float* d; unsigned long long* clk;
    (void)hipMalloc(&d, 4); (void)hipMalloc(&clk, 16);
    for (int rep = 0; rep < 2; ++rep) {
        const int iters = rep == 0 ? 300 : 30000;
        printf("-- %s launch\n", rep == 0 ? "short (~0.2 ms)" : "long (~20 ms)");
        run<0>("zero operands", iters, d, clk);
        run<1>("one constant per operand", iters, d, clk);
        run<2>("random bf16 in [-1, 1)", iters, d, clk);
    }
    return 0;
}
