// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16, bf16/f16 with fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/apadapter_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define APAD_WAVE 64

// Element-type traits: DT is APAD_BF16 or APAD_F16.
template <int DT> struct ET;
template <> struct ET<APAD_BF16> {
    using elem = __bf16;
    using v8 = bf16x8_t;
    using v4 = bf16x4_t;
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct ET<APAD_F16> {
    using elem = _Float16;
    using v8 = f16x8_t;
    using v4 = f16x4_t;
    static __device__ __forceinline__ f32x16 mfma32(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <int DT> __device__ __forceinline__ typename ET<DT>::v8 as_v8(uint4 u) {
    return __builtin_bit_cast(typename ET<DT>::v8, u);
}
template <int DT> __device__ __forceinline__ uint4 as_u4(typename ET<DT>::v8 v) {
    return __builtin_bit_cast(uint4, v);
}
template <int DT> __device__ __forceinline__ float ld_elem(const void* p, int64_t i) {
    return (float)reinterpret_cast<const typename ET<DT>::elem*>(p)[i];
}
template <int DT> __device__ __forceinline__ void st_elem(void* p, int64_t i, float v) {
    reinterpret_cast<typename ET<DT>::elem*>(p)[i] = (typename ET<DT>::elem)v;
}
// 8 floats <-> one 16-byte vector of elements
template <int DT> __device__ __forceinline__ void unpack8(uint4 u, float* f) {
    typename ET<DT>::v8 v = as_v8<DT>(u);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <int DT> __device__ __forceinline__ uint4 pack8(const float* f) {
    typename ET<DT>::v8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (typename ET<DT>::elem)f[i];
    return as_u4<DT>(v);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// host-side error plumbing (capi.cpp)
void apad_set_error(const char* fmt, ...);
#define APAD_CHECK(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            apad_set_error(__VA_ARGS__); \
            return -1;                   \
        }                                \
    } while (0)
int apad_check_launch(const char* what);
