// On-box calibration of the dense bf16 MFMA rate: NCHAIN independent v_mfma_f32_32x32x16_bf16 accumulator chains per
// wave, no memory traffic.  usage: mfma_peak   (prints TF/s for 1, 2 waves per SIMD and 1..4 chains)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NCHAIN> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[NCHAIN];
    for (int c = 0; c < NCHAIN; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NCHAIN; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NCHAIN> void run(int blocks_per_cu, float* d) {
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCHAIN>, dim3(blocks), dim3(256), 0, 0, d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<NCHAIN>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 /*waves*/ * iters * 4 * NCHAIN * 2.0 * 32 * 32 * 16;
    printf("chains=%d waves/SIMD=%d: %.1f TF/s  (%.2f ms)  -> %.1f cycles/MFMA at 2.4 GHz\n", NCHAIN, blocks_per_cu, flops / ms / 1e9, ms,
           ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * 4 * NCHAIN));
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    run<1>(1, d); run<2>(1, d); run<4>(1, d); run<1>(2, d); run<2>(2, d); run<4>(2, d);
    return 0;
}
