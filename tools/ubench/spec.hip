// Wave-specialisation microbenchmark for one gfx950 CU: waves 0..3 (one per SIMD) run an MFMA stream (48 MFMA 32x32x16 bf16 + 24 ds_read_b128 + 8
// ds_write_b128 per iteration), waves 4..7 a VALU stream (NV v_fma_f32 + NT v_exp_f32 per iteration), with / without a workgroup barrier per
// iteration; cycles per iteration of each stream alone and together.   build: hipcc --offload-arch=gfx950 -O3 -o exp/spec tools/ubench/spec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float V16 __attribute__((ext_vector_type(16)));
typedef short B8 __attribute__((ext_vector_type(8)));
typedef uint32_t U4 __attribute__((ext_vector_type(4)));

template <int NV, int NT, bool BAR, int MODE>  // MODE 0: both streams, 1: MFMA waves only (others exit), 2: VALU waves only
__global__ __launch_bounds__(512) void k(uint64_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[65536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t laddr = (uint32_t)(uintptr_t)sm + lane * 16;
    uint64_t t0 = 0, t1 = 0;
    float sink = 0;
    if (wave < 4) {
        if (MODE == 2 && !BAR) return;
        V16 acc0 = {0}, acc1 = {0};
        B8 b0, b1;
        for (int i = 0; i < 8; ++i) { b0[i] = (short)(threadIdx.x + i); b1[i] = (short)(threadIdx.x * 3 + i); }
        asm volatile("" : "+a"(b0), "+a"(b1));
        asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if (BAR) __builtin_amdgcn_s_barrier();
            if (MODE != 2) {
                U4 f[2];
                asm volatile("ds_read_b128 %0, %1" : "=v"(f[0]) : "v"(laddr));
#pragma unroll
                for (int s = 0; s < 24; ++s) {
                    if (s < 23) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[(s + 1) & 1]) : "v"(laddr), "n"(1024));
                    if (s < 23) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    B8 a = __builtin_bit_cast(B8, f[s & 1]);
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "a"(b0));
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "a"(b1));
                }
                asm volatile("s_nop 7\n s_nop 7");
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    U4 w0 = {__float_as_uint(acc0[4 * q]), __float_as_uint(acc0[4 * q + 1]), __float_as_uint(acc0[4 * q + 2]), __float_as_uint(acc0[4 * q + 3])};
                    U4 w1 = {__float_as_uint(acc1[4 * q]), __float_as_uint(acc1[4 * q + 1]), __float_as_uint(acc1[4 * q + 2]), __float_as_uint(acc1[4 * q + 3])};
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(laddr), "v"(w0), "n"(32768) : "memory");
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(laddr), "v"(w1), "n"(40960) : "memory");
                }
            }
        }
        asm volatile("s_nop 7\n s_nop 7\n s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int i = 0; i < 16; ++i) sink += acc0[i] + acc1[i];
    } else {
        if (MODE == 1 && !BAR) return;
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
        float c1 = 1.0001f, c2 = 0.5f;
        U4 d[8];
        asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
            if (BAR) __builtin_amdgcn_s_barrier();
            if (MODE != 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[q]) : "v"(laddr), "n"(32768));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 7]) : "v"(c1), "v"(c2));
                    if (j < NT) asm volatile("v_exp_f32 %0, %0" : "+v"(x[(j + 4) & 7]));
                }
            }
        }
        asm volatile("s_nop 7\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int i = 0; i < 8; ++i) sink += x[i] + __uint_as_float(d[i][0]);
    }
    if (sink == 1234.5678f) out[1000000] = 1;
    if (lane == 0 && (wave == 0 || wave == 4)) out[blockIdx.x * 2 + (wave >> 2)] = t1 - t0;
}

template <int NV, int NT, bool BAR, int MODE>
void run(uint64_t* dout, const char* name, int iters = 500) {
    const int nb = 256;
    hipMemset(dout, 0, nb * 16);
    k<NV, NT, BAR, MODE><<<nb, 512>>>(dout, iters);
    k<NV, NT, BAR, MODE><<<nb, 512>>>(dout, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nb * 2);
    hipMemcpy(h.data(), dout, nb * 16, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < nb; ++i) { a += (double)h[2 * i]; b += (double)h[2 * i + 1]; }
    printf("NV=%3d NT=%2d bar=%d %-10s: MFMA wave %7.0f  VALU wave %7.0f cycles/iteration\n", NV, NT, (int)BAR, name, a / nb / iters, b / nb / iters);
}

int main() {
    uint64_t* dout;
    hipMalloc(&dout, 8 * 1000016);
#define SET(NV, NT) run<NV, NT, false, 1>(dout, "mfma only"); run<NV, NT, false, 2>(dout, "valu only"); run<NV, NT, false, 0>(dout, "both"); run<NV, NT, true, 0>(dout, "both+bar");
    SET(200, 32) SET(250, 32) SET(300, 32) SET(250, 0)
    return 0;
}
