// Write-bandwidth probe: how fast can 131 MB be written with (A) contiguous 16 B/lane stores, (B) 32 rows x 64 B
// row-strided segments per wave-instruction pair (the row-panel GEMM epilogue pattern), (C) 128 B segments,
// (D) = B with non-temporal stores, (E) each wave owns 32 full rows and sweeps them left to right.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int ROWS = 64000, COLS = 1024;  // bf16 elements -> 2048 B rows, 131 MB
__global__ void kA(u32x4* o, long n) { long i = (long)blockIdx.x * 256 + threadIdx.x; u32x4 v = {1u, 2u, 3u, (unsigned)i}; if (i < n) o[i] = v; }
template <int SEG, bool NT>  // SEG bytes per row per flush; block = 4 waves x 32 rows; grid.y splits the columns in 2
__global__ void kB(unsigned char* o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
    const int cpr = SEG / 16;
    const int c_begin = blockIdx.y * 1024, c_end = c_begin + 1024;  // bytes within the row
    for (int c = c_begin; c < c_end; c += SEG)
        for (int idx = lane; idx < 32 * cpr; idx += 64) {
            const int row = idx / cpr, ch = idx % cpr;
            u32x4 v = {1u, 2u, (unsigned)c, (unsigned)idx};
            u32x4* p = (u32x4*)(o + (r0 + row) * 2048 + c + ch * 16);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
}
__global__ void kE(unsigned char* o) {  // wave owns 2 rows at a time, writes them fully (2 x 1 KB per instruction pair)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = ((long)blockIdx.x * 4 + wave) * 32;
    for (int r = 0; r < 32; ++r)
        for (int c = lane * 16; c < 2048; c += 1024) {
            u32x4 v = {1u, 2u, (unsigned)c, (unsigned)r};
            *(u32x4*)(o + (r0 + r) * 2048 + c) = v;
        }
}
template <class F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
    unsigned char* o; const long bytes = (long)ROWS * COLS * 2; hipMalloc(&o, bytes);
    auto rep = [&](const char* n, float ms) { printf("%-44s %8.1f us  %7.1f GB/s\n", n, ms * 1e3, bytes / ms / 1e6); };
    rep("A contiguous 16B/lane", timeit([&] { kA<<<dim3((unsigned)(bytes / 16 / 256)), 256>>>((u32x4*)o, bytes / 16); }));
    rep("B 32 rows x 64B segments", timeit([&] { kB<64, false><<<dim3(ROWS / 128, 2), 256>>>(o); }));
    rep("C 32 rows x 128B segments", timeit([&] { kB<128, false><<<dim3(ROWS / 128, 2), 256>>>(o); }));
    rep("C2 32 rows x 256B segments", timeit([&] { kB<256, false><<<dim3(ROWS / 128, 2), 256>>>(o); }));
    rep("D 64B segments, non-temporal", timeit([&] { kB<64, true><<<dim3(ROWS / 128, 2), 256>>>(o); }));
    rep("D2 128B segments, non-temporal", timeit([&] { kB<128, true><<<dim3(ROWS / 128, 2), 256>>>(o); }));
    rep("E wave sweeps whole rows", timeit([&] { kE<<<dim3(ROWS / 128), 256>>>(o); }));
    return 0;
}
