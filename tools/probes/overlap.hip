// Does ONE wave overlap its own MFMAs with independent VALU work that follows them in program order?
// Per iteration: NM MFMAs (32 cycles each on the matrix pipe) and NV v_exp_f32 + v_fma (VALU).  Variants: MFMA only, VALU
// only, clustered (all MFMAs then all VALU), interleaved (1 MFMA : NV/NM VALU).  One wave per SIMD (256 threads / CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {  // clustered: 8 MFMAs, then 32 (exp + fma)
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c & 3], 0, 0, 0);
        }
        if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.5f + 0.25f;
        }
        if (MODE == 3) {  // interleaved 1 MFMA : 4 (exp + fma), pinned
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[c * 4 + i] = __builtin_amdgcn_exp2f(v[c * 4 + i]) * 0.5f + 0.25f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 5) {  // VALU only, transcendentals clustered: 32 exps back to back, then 32 fma
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = v[i] * 0.5f + 0.25f;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 6) {  // VALU only, pinned 1 exp : 1 fma of ANOTHER element (independent)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                v[i] = __builtin_amdgcn_exp2f(v[i]);
                v[(i + 16) & 31] = v[(i + 16) & 31] * 0.5f + 0.25f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 7) {  // VALU only, 1 exp : 3 plain VALU
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                v[i] = __builtin_amdgcn_exp2f(v[i]);
                v[(i + 16) & 31] = v[(i + 16) & 31] * 0.5f + 0.25f;
                v[(i + 8) & 31] = v[(i + 8) & 31] * 0.75f + 0.125f;
                v[(i + 24) & 31] = v[(i + 24) & 31] * 0.25f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE == 4) {  // VALU first, then MFMAs (pinned)
#pragma unroll
            for (int i = 0; i < 32; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]) * 0.5f + 0.25f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 32; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

template <int MODE> float run(float* d, int wpb) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wpb), dim3(256), 0, 0, d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wpb), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3f * 2.4e9f / iters / wpb;  // cycles per iteration per wave-slot at 2.4 GHz
}

int main() {
    float* d; (void)hipMalloc(&d, 4);
    for (int wpb = 1; wpb <= 2; ++wpb) {
        printf("waves/SIMD=%d  (8 MFMA = 256 pipe cycles; 32 exp + 32 fma)\n", wpb);
        printf("  MFMA only            %7.0f cycles/iter\n", run<0>(d, wpb));
        printf("  VALU only            %7.0f\n", run<1>(d, wpb));
        printf("  MFMAs then VALU      %7.0f\n", run<2>(d, wpb));
        printf("  interleaved 1:4      %7.0f\n", run<3>(d, wpb));
        printf("  VALU then MFMAs      %7.0f\n", run<4>(d, wpb));
        printf("  VALU: 32 exp clustered, then 32 fma   %7.0f\n", run<5>(d, wpb));
        printf("  VALU: exp : fma 1:1 pinned            %7.0f\n", run<6>(d, wpb));
        printf("  VALU: exp : 3 fma pinned (32 + 96)    %7.0f\n", run<7>(d, wpb));
    }
    return 0;
}
