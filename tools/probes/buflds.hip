// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS to LDS for lanes whose offset is out of range (raw buffer, stride 0)?
// (the implicit-GEMM convolution relies on it for the zero padding: tools/probes/buflds -> "oob lanes wrote zero: yes")
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint8_t* a, uint32_t bytes, uint8_t* out, const uint32_t* offs, int soff) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    reinterpret_cast<uint4*>(smem)[threadIdx.x] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a), 0, bytes, 0x00020000);
    uint32_t voff = offs[threadIdx.x];
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem), 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    reinterpret_cast<uint4*>(out)[threadIdx.x] = reinterpret_cast<uint4*>(smem)[threadIdx.x];
}
int main() {
    const uint32_t bytes = 1 << 20;
    std::vector<uint32_t> h(bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i * 4 + 1;  // word value = its byte offset + 1
    uint8_t *a, *out; uint32_t* offs;
    hipMalloc(&a, 2 * bytes); hipMalloc(&out, 1024); hipMalloc(&offs, 256);
    hipMemcpy(a, h.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(a + bytes, h.data(), bytes, hipMemcpyHostToDevice);  // memory behind the declared range is mapped and non-zero
    for (int soff : {0, 4096}) {
        std::vector<uint32_t> o(64);
        for (int i = 0; i < 64; ++i) o[i] = (i % 3 == 0) ? 0x80000000u : (i % 3 == 1 ? bytes - 16 + ((i & 4) ? 0 : 16) : (uint32_t)i * 64);
        hipMemcpy(offs, o.data(), 256, hipMemcpyHostToDevice);
        k<<<1, 64, 1024>>>(a, bytes, out, offs, soff);
        std::vector<uint32_t> r(256);
        hipMemcpy(r.data(), out, 1024, hipMemcpyDeviceToHost);
        int oob_zero = 0, oob_n = 0, ok = 0, okn = 0, edge_zero = 0, edge_n = 0, edge_data = 0;
        for (int i = 0; i < 64; ++i) {
            const uint32_t v = r[i * 4];
            if (o[i] == 0x80000000u) { ++oob_n; oob_zero += v == 0; }
            else if (o[i] + soff + 16 <= bytes) { ++okn; ok += v == o[i] + soff + 1; }
            else { ++edge_n; edge_zero += v == 0; edge_data += v == ((o[i] + soff) % bytes) + 1; }
        }
        printf("soffset %d: oob lanes wrote zero: %d/%d; in-range lanes correct: %d/%d; lanes past the end (voff+soff+16 > size): zero %d / data %d of %d\n",
               soff, oob_zero, oob_n, ok, okn, edge_zero, edge_data, edge_n);
    }
    return 0;
}
