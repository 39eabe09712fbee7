// Issue cost of the transcendental instructions on gfx950, one wave per SIMD, no memory traffic: N dependent-free
// instructions per iteration on independent registers, cycles per instruction from s_memtime.
// build: hipcc --offload-arch=gfx950 -O3 -o exp_rate exp_rate.hip ; run: ./exp_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = -0.001f * (threadIdx.x + i);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[i]));
            if (MODE == 1) asm volatile("v_exp_legacy_f32_e32 %0, %0" : "+v"(v[i]));
            if (MODE == 2) asm volatile("v_exp_f16_e32 %0, %0" : "+v"(v[i]));
            if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
            if (MODE == 4) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(v[i]));
            if (MODE == 5) asm volatile("v_log_f32_e32 %0, %0" : "+v"(v[i]));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, float* out, long long* cyc) {
    const int iters = 4000, blocks = 1024;  // 256 CUs x 4 SIMDs, one wave each
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // s_memtime ticks at 100 MHz: report wall time per instruction and convert with the measured kernel time
    printf("%-22s %8.3f ms  -> %6.2f ns per wave-instruction (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / (iters * 16.0),
           ms * 1e6 / (iters * 16.0) * 2.4);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 1024 * 8);
    run<3>("v_fma_f32", out, cyc);
    run<0>("v_exp_f32", out, cyc);
    run<1>("v_exp_legacy_f32", out, cyc);
    run<2>("v_exp_f16", out, cyc);
    run<4>("v_rcp_f32", out, cyc);
    run<5>("v_log_f32", out, cyc);
    return 0;
}
