// LDS fill throughput per CU from L2-resident memory: (0) the LDS-DMA (buffer_load_dwordx4 ... lds) against (1) buffer_load_dwordx4 into registers
// + ds_write_b128, eight 1 KB pieces per round, two register sets (the next round's loads are in flight while this round is written).
//   build: hipcc --offload-arch=gfx950 -O3 -o exp/fill tools/ubench/fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const uint8_t* src, uint64_t* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 1 << 24, 0x00020000);
    const uint32_t voff = lane * 16;
    const uint32_t la = (uint32_t)(uintptr_t)(lds_ptr)smem + wave * 8192 + lane * 16;
    uint64_t t0, t1;
    U4 a[8], b[8];
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int base = (blockIdx.x % 32) * 262144 + wave * 8192;
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
            const int so = base + (it & 1) * 131072;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                lds_ptr lp = (lds_ptr)(smem + wave * 8192 + g * 4096);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so + g * 4096, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so + g * 4096, 1024, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so + g * 4096, 2048, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so + g * 4096, 3072, 0);
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, base + q * 1024, 0);
        for (int it = 0; it < iters; it += 2) {
            const int so = base + 131072;
#pragma unroll
            for (int q = 0; q < 8; ++q) b[q] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, so + q * 1024, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(la), "v"(a[q]), "n"(0) : "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, base + q * 1024, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(la), "v"(b[q]), "n"(1024) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    uint32_t s = 0;
    if (MODE == 1) for (int i = 0; i < 8; ++i) s += a[i][0] + b[i][0];
    if (s == 0x12345678u) out[100000] = 1;
    if (lane == 0) { out[(blockIdx.x * 16 + wave) * 2] = t0; out[(blockIdx.x * 16 + wave) * 2 + 1] = t1; }
}
template <int MODE> void run(const uint8_t* src, uint64_t* dout, int waves) {
    const int nb = 256, iters = 400;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    k<MODE><<<nb, waves * 64, 128 * 1024>>>(src, dout, iters);
    k<MODE><<<nb, waves * 64, 128 * 1024>>>(src, dout, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nb * 32);
    hipMemcpy(h.data(), dout, nb * 32 * 8, hipMemcpyDeviceToHost);
    double tot = 0;
    for (int b = 0; b < nb; ++b) {
        uint64_t lo = ~0ull, hi = 0;
        for (int w = 0; w < waves; ++w) { lo = std::min(lo, h[(b * 16 + w) * 2]); hi = std::max(hi, h[(b * 16 + w) * 2 + 1]); }
        tot += (double)(hi - lo);
    }
    printf("%-42s %2d waves/CU: %6.1f bytes / cycle / CU\n", MODE == 0 ? "LDS-DMA" : "buffer_load_dwordx4 + ds_write_b128", waves, (double)waves * iters * 8192 / (tot / nb));
}
int main() {
    uint8_t* src; uint64_t* dout;
    hipMalloc(&src, 1 << 24); hipMemset(src, 1, 1 << 24);
    hipMalloc(&dout, 8 * 100016);
    for (int w : {4, 8, 16}) { run<0>(src, dout, w); run<1>(src, dout, w); }
    return 0;
}
