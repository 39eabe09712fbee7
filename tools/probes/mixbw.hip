// Ceiling for the to_out / proj_out traffic mix: out[i] = a[i] + b[i] over 64000 x 256 16-bit values (two 32.8 MB reads + one 32.8 MB write, 16 B per lane,
// contiguous), nothing else.  build: hipcc --offload-arch=gfx950 -O3 -o mixbw mixbw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const u32x4* a, const u32x4* b, u32x4* o, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { u32x4 x = a[i], y = b[i]; o[i] = x + y; }
}
__global__ void k4(const u32x4* a, const u32x4* b, u32x4* o, long n) {  // 4 chunks per thread, loads first
    long i = ((long)blockIdx.x * 256 + threadIdx.x);
    const long s = (long)gridDim.x * 256;
    u32x4 x[4], y[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) if (i + j * s < n) { x[j] = a[i + j * s]; y[j] = b[i + j * s]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) if (i + j * s < n) o[i + j * s] = x[j] + y[j];
}
int main() {
    const long bytes = 64000L * 256 * 2, n = bytes / 16;
    u32x4 *a, *b, *o; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, bytes); hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        auto run = [&] { if (v == 0) k<<<dim3((unsigned)((n + 255) / 256)), 256>>>(a, b, o, n); else k4<<<dim3((unsigned)((n / 4 + 255) / 256)), 256>>>(a, b, o, n); };
        run(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 50; ++i) run(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 50;
        printf("%s: %.1f us  %.2f TB/s (2 reads + 1 write of 32.8 MB)\n", v ? "4 chunks per thread" : "1 chunk per thread", ms * 1e3, 3.0 * bytes / ms / 1e9);
    }
    return 0;
}
