// Issue cost of the LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per instruction) against a plain 16-byte global load into registers, per wave:
// N instructions back to back from L2-resident memory, s_memtime before the first and after the last ISSUE (not the completion), then the drain.
//   build: hipcc --offload-arch=gfx950 -O3 -o exp/dmaissue tools/ubench/dmaissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int MODE, int WAVES>  // 0: LDS-DMA with immediate offsets (one M0 per 4), 1: LDS-DMA, one M0 write each, 2: global_load_dwordx4 to VGPRs, 3: mode 0 interleaved with 8 v_fma
__global__ __launch_bounds__(256) void k(const uint8_t* src, uint64_t* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 1 << 24, 0x00020000);
    const uint32_t voff = lane * 16;
    uint64_t t0, t1, t2;
    U4 d[8];
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = lane + i;
    uint64_t issue = 0, drain = 0;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int so = (blockIdx.x % 64) * 65536 + wave * 24576 + g * 4096;
            if (MODE == 0 || MODE == 3) {
                lds_ptr lp = (lds_ptr)(smem + wave * 24576 + g * 4096);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so, 0, 0);
                if (MODE == 3) { for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j])); }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so, 1024, 0);
                if (MODE == 3) { for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j])); }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so, 2048, 0);
                if (MODE == 3) { for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j])); }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so, 3072, 0);
                if (MODE == 3) { for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j])); }
            } else if (MODE == 1) {
                for (int q = 0; q < 4; ++q) {
                    lds_ptr lp = (lds_ptr)(smem + wave * 24576 + g * 4096 + q * 1024);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lp, 16, voff, so + q * 1024, 0, 0);
                }
            } else {
                for (int q = 0; q < 4; ++q) d[(g * 4 + q) & 7] = *reinterpret_cast<const U4*>(src + so + q * 1024 + voff);
            }
        }
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        if (it > 0) { issue += t1 - t0; drain += t2 - t1; }
    }
    uint32_t s = 0;
    if (MODE == 2) for (int i = 0; i < 8; ++i) s += d[i][0];
    for (int i = 0; i < 8; ++i) s += (uint32_t)x[i];
    if (s == 0x12345678u) out[100000] = 1;
    if (lane == 0 && wave == 0) { out[blockIdx.x * 2] = issue; out[blockIdx.x * 2 + 1] = drain; }
}
template <int MODE, int WAVES> void run(const uint8_t* src, uint64_t* dout, const char* name) {
    const int nb = 256, iters = 200;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    k<MODE, WAVES><<<nb, WAVES * 64, 100 * 1024>>>(src, dout, iters);
    hipDeviceSynchronize();
    std::vector<uint64_t> h(nb * 2);
    hipMemcpy(h.data(), dout, nb * 16, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < nb; ++i) { a += (double)h[2 * i]; b += (double)h[2 * i + 1]; }
    printf("%-44s waves/CU=%d: issue %6.1f ticks per 1 KB instruction, drain %6.0f ticks after the last of 24\n", name, WAVES, a / nb / (iters - 1) / 24, b / nb / (iters - 1));
}
int main() {
    uint8_t* src; uint64_t* dout;
    hipMalloc(&src, 1 << 24); hipMemset(src, 1, 1 << 24);
    hipMalloc(&dout, 8 * 100016);
    run<0, 1>(src, dout, "LDS-DMA, immediate offsets");  run<0, 4>(src, dout, "LDS-DMA, immediate offsets");
    run<1, 1>(src, dout, "LDS-DMA, one M0 each");        run<1, 4>(src, dout, "LDS-DMA, one M0 each");
    run<2, 1>(src, dout, "global_load_dwordx4 -> VGPR"); run<2, 4>(src, dout, "global_load_dwordx4 -> VGPR");
    run<3, 1>(src, dout, "LDS-DMA + 8 v_fma behind each");  run<3, 4>(src, dout, "LDS-DMA + 8 v_fma behind each");
    return 0;
}
