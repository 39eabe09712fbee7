// Pieces shared by the 64-token register-block kernels (mlp3.hip: the whole feed-forward at C = 256; geglu3.hip: LayerNorm + GEGLU projection at
// C = 384): inline-asm LDS fragment reads + counted waits beside an LDS-DMA ring, the GEGLU arithmetic in placeable phases, VGPR-accumulator MFMAs.
#pragma once
#include "rp_shared.h"

namespace {

typedef __attribute__((address_space(3))) void* m3_lds_ptr;

template <int OFF> __device__ __forceinline__ void m3_read(u32x4& d, uint32_t a) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(a), "n"(OFF));
}
template <int F0> __device__ __forceinline__ void m3_read2(u32x4 (&f)[2], uint32_t a) {
    m3_read<F0 * 1024>(f[0], a);
    m3_read<(F0 + 1) * 1024>(f[1], a);
}
template <int N> __device__ __forceinline__ void m3_wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#define M3_FENCE() asm volatile("" ::: "memory")

// GEGLU of two hidden units (value v*, gate g*) in four phases of ~8 vector instructions, so that the phases can be placed between the MFMAs of a
// step by hand (one wave per SIMD: an MFMA covers the few vector instructions issued right behind it, nothing else does).  gelu_erf_2's arithmetic
// (A&S 7.1.26) in the same operation order -> the same bits as mlp_kernel / mlp2_kernel.
// (a packed-math form of these phases -- v_pk_mul / v_pk_fma, 22 instead of 31 instructions per pair -- measured SLOWER, 113.0 -> 122.5 us: the
//  dependent packed operations need hazard s_nops and cost more issue time beside the MFMAs than the scalar ones they replace)
// PRE: the value half arrives pre-multiplied by 0.5 (apad_mlp_pack / apad_geglu_pack halve the value rows of W1 and b1 for bf16 -- exact: a power of
// two commutes with every rounding on the way), so gelu(g) v = (g + |g| erf) (v / 2) needs no `0.5 g`; f16 keeps the multiply (halving a weight
// below 2^-14 would lose its last bit).  Either way the same bits as gelu_erf_2's  (h + |h| erf) v,  h = g / 2.
template <bool PRE> struct M3GegluT {
    float g0, g1, t0, t1, q0, q1, e0, e1, p0, p1, r0, r1;
    __device__ __forceinline__ void ph1(float ga, float gb) {
        g0 = ga; g1 = gb;
        const float z0 = g0 * APAD_GELU_K1, z1 = g1 * APAD_GELU_K1;
        t0 = __builtin_amdgcn_rcpf(fmaf(fabsf(z0), APAD_GELU_P1, 1.0f));
        t1 = __builtin_amdgcn_rcpf(fmaf(fabsf(z1), APAD_GELU_P1, 1.0f));
        q0 = z0 * z0; q1 = z1 * z1;
        asm volatile("" : "+v"(t0), "+v"(t1), "+v"(q0), "+v"(q1));  // (anchors: the phase is computed HERE, between the MFMAs around it)
    }
    __device__ __forceinline__ void ph2() {
        e0 = __builtin_amdgcn_exp2f(-q0);
        e1 = __builtin_amdgcn_exp2f(-q1);
        p0 = fmaf(fmaf(t0, 1.061405429f, -1.453152027f), t0, 1.421413741f);
        p1 = fmaf(fmaf(t1, 1.061405429f, -1.453152027f), t1, 1.421413741f);
        asm volatile("" : "+v"(e0), "+v"(e1), "+v"(p0), "+v"(p1));
    }
    __device__ __forceinline__ void ph3() {
        p0 = fmaf(fmaf(p0, t0, -0.284496736f), t0, 0.254829592f);
        p1 = fmaf(fmaf(p1, t1, -0.284496736f), t1, 0.254829592f);
        r0 = fmaf(p0 * t0, -e0, 1.0f);
        r1 = fmaf(p1 * t1, -e1, 1.0f);
        asm volatile("" : "+v"(r0), "+v"(r1));
    }
    template <typename V8, typename EL> __device__ __forceinline__ void ph4(float v0, float v1, V8& hn, int r) {
        // (one value after the other: the SLP vectoriser pairs them into v_pk_* otherwise, each followed by a hazard s_nop)
        float h0 = g0, h1 = g1;
        if (!PRE) {
            h0 *= 0.5f;
            asm volatile("" : "+v"(h0));
            h1 *= 0.5f;
            asm volatile("" : "+v"(h1));
        }
        float u0 = fmaf(fabsf(h0), r0, h0);
        asm volatile("" : "+v"(u0));
        float u1 = fmaf(fabsf(h1), r1, h1);
        asm volatile("" : "+v"(u1));
        u0 *= v0;
        asm volatile("" : "+v"(u0));
        u1 *= v1;
        asm volatile("" : "+v"(u1));  // (fp32 product, then one rounding -- never a fused v_fma_mix: bit-equal to mlp_kernel / mlp2_kernel)
        hn[r] = (EL)u0;
        hn[r + 1] = (EL)u1;
    }
};
template <int DT> using M3Geglu = M3GegluT<DT == APAD_BF16>;
template <int DT> constexpr float m3_value_scale() { return DT == APAD_BF16 ? 0.5f : 1.0f; }

// gemm1's MFMAs are inline asm with VGPR accumulators: the compiler's MFMAs of this function are the AGPR form (the 256 output accumulators fill the
// AGPR file), and an AGPR-form accumulator for gemm1 would have to be copied out through v_accvgpr_read for the GEGLU arithmetic (and, with 320
// accumulator registers asked of a 256-entry file, shuffled between AGPR ranges: measured in the ISA, 8 copies per MFMA).  Hazards the compiler would
// have covered: the accumulators are read by vector instructions only an LDS round trip (the next iteration's fragment wait) after the last MFMA
// that writes them; the first MFMA's C operand (b1, straight from ds_read_b128) sits behind an explicit wait + s_nop.
template <int DT> struct M3Asm;
template <> struct M3Asm<APAD_BF16> {
    template <typename V8> static __device__ __forceinline__ void first(f32x16& d, const V8& a, const V8& b, const f32x16& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    }
    template <typename V8> static __device__ __forceinline__ void acc(f32x16& d, const V8& a, const V8& b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
};
template <> struct M3Asm<APAD_F16> {
    template <typename V8> static __device__ __forceinline__ void first(f32x16& d, const V8& a, const V8& b, const f32x16& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    }
    template <typename V8> static __device__ __forceinline__ void acc(f32x16& d, const V8& a, const V8& b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    }
};
#define M3_PIN() __builtin_amdgcn_sched_barrier(0)


}  // namespace
