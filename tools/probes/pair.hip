// Two waves on ONE SIMD with DIFFERENT instruction streams: does a VALU-only wave run beside an MFMA-only wave for free?
// 512-thread workgroups (waves w and w+4 share a SIMD), one per CU.  Role per half: 0 = idle (exits), 1 = MFMA only (8 per
// iteration, 4 accumulator chains), 2 = VALU only (32 x (v_exp_f32 + v_fma_f32)), 3 = plain VALU only (64 v_fma_f32),
// 4 = one dependent MFMA chain.  Prints wall cycles per iteration of the whole workgroup (max over both halves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROLE> __device__ __forceinline__ float body(int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if (ROLE == 1) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c & 3], 0, 0, 0);
        }
        if (ROLE == 4) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
        }
        if (ROLE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.5f + 0.25f;
        }
        if (ROLE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { v[i] = v[i] * 0.5f + 0.25f; v[i] = v[i] * 0.75f + 0.125f; }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 32; ++i) s += v[i];
    return s;
}

template <int RA, int RB> __global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if (wave < 4) { if (RA) s = body<RA>(iters); } else { if (RB) s = body<RB>(iters); }
    if (s == 12345.678f) out[0] = s;
}

template <int RA, int RB> float run(float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / iters;  // cycles per iteration at 2.4 GHz
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    printf("per iteration: MFMA role = 8 x 32x32x16 (256 pipe cycles); VALU role = 32 exp + 32 fma; plain = 64 fma\n");
    printf("  MFMA alone                 %6.0f\n", run<1, 0>(d));
    printf("  dependent MFMA chain alone %6.0f\n", run<4, 0>(d));
    printf("  VALU(exp) alone            %6.0f\n", run<2, 0>(d));
    printf("  plain VALU alone           %6.0f\n", run<3, 0>(d));
    printf("  MFMA + MFMA                %6.0f\n", run<1, 1>(d));
    printf("  VALU(exp) + VALU(exp)      %6.0f\n", run<2, 2>(d));
    printf("  plain + plain              %6.0f\n", run<3, 3>(d));
    printf("  MFMA + VALU(exp)           %6.0f\n", run<1, 2>(d));
    printf("  VALU(exp) + MFMA           %6.0f\n", run<2, 1>(d));
    printf("  MFMA + plain VALU          %6.0f\n", run<1, 3>(d));
    printf("  dep. MFMA chain + VALU(exp)%6.0f\n", run<4, 2>(d));
    printf("  dep. MFMA chain + plain    %6.0f\n", run<4, 3>(d));
    return 0;
}
