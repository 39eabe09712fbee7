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

// fp32 precision mode: only the element type (the element-wise kernels are templated on it); the MFMA kernels of this
// mode live in f32_ops.hip
template <> struct ET<APAD_F32> {
    using elem = float;
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

// Exchange between the two 32-lane halves of a wave (lane l <-> lane l ^ 32), the only cross-lane traffic of the
// transposed-score kernels.  gfx950's v_permlane32_swap does it in one VALU instruction; __shfl_xor(v, 32) lowers to
// ds_bpermute_b32, i.e. an LDS round trip plus an lgkmcnt wait in the middle of the softmax.
__device__ __forceinline__ void half_pair(float v, float& lo, float& hi) {
    const unsigned a = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    lo = __uint_as_float(r[0]);  // the lower half's value, in every lane
    hi = __uint_as_float(r[1]);  // the upper half's value, in every lane
}
__device__ __forceinline__ float half_max(float v) { float lo, hi; half_pair(v, lo, hi); return fmaxf(lo, hi); }
__device__ __forceinline__ float half_sum(float v) { float lo, hi; half_pair(v, lo, hi); return lo + hi; }
__device__ __forceinline__ float half_lo(float v) { float lo, hi; half_pair(v, lo, hi); return lo; }

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// erf via Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32-exact for a bf16/f16 result): one rcp, one exp2 and
// five FMAs instead of libm's erff (~40 VALU instructions) -- the GEGLU epilogue evaluates this for every element of the
// 4C-wide feed-forward activation, where erff made the epilogue cost more than the MFMAs feeding it.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = fmaf(-p * t, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// Two GELUs at once on packed fp32 math (v_pk_mul/fma_f32 process two lanes-worth per instruction; only rcp / exp2 stay
// scalar).  A&S 7.1.26 with the argument scaled ONCE: z' = x sqrt(log2(e) / 2), so that exp(-x^2 / 2) = exp2(-z'^2) (the negation is a source
// modifier) and the rational argument is |z'| p' with p' = p / sqrt(log2 e) -- two multiplies per value fewer than scaling x / sqrt 2 and
// z^2 log2(e) separately.  Every GEGLU kernel of the library evaluates THIS operation order (the phased form: M3Geglu, mlp3_shared.h), which
// is what makes a row's result independent of the kernel its batch size selects.
typedef float apad_f32x2 __attribute__((ext_vector_type(2)));
constexpr float APAD_GELU_K1 = 0.84932180028801904272f;  // sqrt(log2(e) / 2)
constexpr float APAD_GELU_P1 = 0.2727374808792225f;       // 0.3275911 / sqrt(log2(e))
__device__ __forceinline__ apad_f32x2 gelu_erf_2(apad_f32x2 x) {
    const apad_f32x2 z = x * APAD_GELU_K1;
    const apad_f32x2 az = {fabsf(z[0]), fabsf(z[1])};
    const apad_f32x2 d = __builtin_elementwise_fma(az, (apad_f32x2){APAD_GELU_P1, APAD_GELU_P1}, (apad_f32x2){1.0f, 1.0f});
    const apad_f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    apad_f32x2 p = __builtin_elementwise_fma(t, (apad_f32x2){1.061405429f, 1.061405429f}, (apad_f32x2){-1.453152027f, -1.453152027f});
    p = __builtin_elementwise_fma(p, t, (apad_f32x2){1.421413741f, 1.421413741f});
    p = __builtin_elementwise_fma(p, t, (apad_f32x2){-0.284496736f, -0.284496736f});
    p = __builtin_elementwise_fma(p, t, (apad_f32x2){0.254829592f, 0.254829592f});
    const apad_f32x2 q = z * z;
    const apad_f32x2 e = {__builtin_amdgcn_exp2f(-q[0]), __builtin_amdgcn_exp2f(-q[1])};
    const apad_f32x2 r = __builtin_elementwise_fma(p * t, -e, (apad_f32x2){1.0f, 1.0f});  // erf(|x| / sqrt 2)
    const apad_f32x2 hx = x * 0.5f;
    // 0.5 x (1 + sign(x) erf|z|) = hx + |hx| * erf|z|
    const apad_f32x2 ahx = {fabsf(hx[0]), fabsf(hx[1])};
    return __builtin_elementwise_fma(ahx, r, hx);
}

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
// dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, DEVICE): *devmask = the devices it has been set on
// (one static word per kernel at the call site); thread-safe, the return code of hipFuncSetAttribute is checked.  0 = ok, -1 = error set
int apad_ensure_dyn_lds(const void* kern, int bytes, unsigned* devmask);
// big-tile GEMM / implicit convolution (cgemm.hip): 1 = not applicable, 0 = launched, < 0 = error
int apad_cgemm_try(const apad_gemm_desc* d, hipStream_t s);
// halo-resident 3x3 convolution (hconv.hip), same convention; needs apad_gemm_desc::w_halo
int apad_hconv_try(const apad_gemm_desc* d, hipStream_t s);
