// apad_geglu_mlp: the whole feed-forward of a BasicTransformerBlock in ONE kernel:
//     out = x + W2 . ( value * gelu(gate) ) + b2,   [value | gate] = W1 . LayerNorm(x) + b1        (diffusers FeedForward/GEGLU)
// The 8C-wide projection and the 4C-wide activation never exist in memory: per 16 hidden units the kernel runs the first GEMM (K = C, x fragments
// resident in registers, LayerNorm applied in registers), forms value * gelu(gate) in registers, and feeds it -- through a 1 KB LDS hand-over between
// the two waves of a pair -- to the second GEMM as its B operand: the C layout of the first MFMA (lane = token, registers = hidden units
// (r & 3) + 8 (r >> 2) + 4 half) IS a valid B-operand layout of the second MFMA as long as the W2 fragment is read with the same hidden-unit
// permutation -- the trick apad_attention uses for P.V.
// This file holds the 128-token-workgroup form from UNPACKED weights (mlp2_kernel): launches below 48 000 rows (the CFG-shared prefix, small
// batches, the training step).  Full-size launches: the 64-token register-block kernel from packed weights (mlp3.hip), bit-equal to this one.
// (Rounds 1-4 also kept a one-wave-per-SIMD form, mlp_kernel: slower at every size, removed in round 5 -- NOTES 4b / 10.)
#include <stdlib.h>
#include "rp_shared.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct MlpP {
    const uint8_t* x;
    const uint8_t* gamma;
    const uint8_t* beta;
    const uint8_t* w1;
    const uint8_t* b1;
    const uint8_t* w2;
    const uint8_t* b2;
    uint8_t* out;
    int64_t M;
    float eps;
};

// ---------------------------------------------------------------------------------------------------------------------
// Two waves per SIMD (round 2).  mlp_kernel keeps a whole 32-token panel's output (128 accumulator registers) in one wave
// and therefore runs ONE wave per SIMD: nothing covers that wave's own LDS / dependency latencies (PMC: parked 28 %, issue
// stalled 29 %).  Here the 8 waves of a 512-thread workgroup form 4 PAIRS; the two waves of a pair share one token panel and
// split the work of every 32-hidden-unit chunk by ROWS: wave `hh` runs the first GEMM and the GEGLU arithmetic of hidden units
// 16hh..16hh+15 of the chunk, hands its 1 KB of activations to its partner through LDS, and accumulates output columns
// 128hh..128hh+127 from BOTH halves.  Per wave: x panel 64 + output accumulators 64 registers -> the kernel fits the 256
// registers of two waves per SIMD.  MFMA count, LDS fragment traffic and GELU work per token are those of mlp_kernel; the
// exchange rides on the per-chunk barrier the weight staging needs anyway (the second GEMM runs one chunk late).
template <int KC> struct Mlp2Cfg {
    static constexpr int C = KC * 16, CT = C / 64, HID = 4 * C, NCHUNK = HID / 32;
    static constexpr int ROWB1 = Cfg<KC>::ROWB;
    static constexpr int W1_BYTES = 64 * ROWB1;    // per half: 16 value rows, 16 gate rows
    static constexpr int ROWB2 = 72;               // 32 hidden units (64 B) + 8 B pad: conflict-free b64 reads
    static constexpr int W2_BYTES = C * ROWB2;
    static constexpr int HX_BYTES = 4 * 2 * 1024;  // [pair][half] x 64 lanes x 16 B
    static constexpr int LDS = 2 * W1_BYTES + 3 * W2_BYTES + 2 * HX_BYTES + (2 * HID + C) * 4;
};

template <int DT, int KC, bool LN>
__global__ __launch_bounds__(512, 1) void mlp2_kernel(MlpP p) {
    using E = ET<DT>;
    using G = Mlp2Cfg<KC>;
    constexpr int NTH = 512;
    constexpr int N1 = 64 * KC * 2 / NTH, N2 = G::C * 4 / NTH, NG = KC / 4;
    static_assert(NG == 4 && G::CT == 4, "C = 256");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int pair = wave >> 1, hh = wave & 1;
    const int64_t mw0 = ((int64_t)blockIdx.x * 4 + pair) * 32;

    uint8_t* const w1s = smem;                                        // 2 stages
    uint8_t* const w2s = smem + 2 * G::W1_BYTES;                      // 3 slots (the second GEMM lags one chunk)
    uint8_t* const hxs = w2s + 3 * G::W2_BYTES;                       // 2 parities of the activation hand-over
    float* const lb1 = reinterpret_cast<float*>(hxs + 2 * G::HX_BYTES);
    float* const lb2 = lb1 + 2 * G::HID;

    u32x4 s1[N1], s2[N2];
    // staging addresses: thread t moves 16-byte chunk (t % 32) of W1-tile rows (t / 32) + 16 i -- i = 0..3 are the value / gate rows of
    // half 0, then of half 1, i.e. compile-time row offsets -- and chunk (t % 4) of W2 rows (t / 4) + 128 i
    static_assert(N1 == 4 && N2 == 2, "C = 256, 512 threads");
    const uint32_t o1 = (uint32_t)((((tid >> 5)) * G::C + (tid & 31) * 8) * 2);
    const uint32_t d1 = (uint32_t)((tid >> 5) * G::ROWB1 + (tid & 31) * 16);
    const uint32_t o2 = (uint32_t)((((int64_t)(tid >> 2)) * G::HID + (tid & 3) * 8) * 2);
    const uint32_t d2 = (uint32_t)((tid >> 2) * G::ROWB2 + (tid & 3) * 16);
    constexpr int64_t SRC1[4] = {0, (int64_t)G::HID * G::C * 2, 16 * G::C * 2, ((int64_t)G::HID + 16) * G::C * 2};
    auto load_w1 = [&](int jc) {
        const uint8_t* base = p.w1 + (int64_t)jc * 32 * G::C * 2 + o1;
#pragma unroll
        for (int i = 0; i < N1; ++i) s1[i] = *reinterpret_cast<const u32x4*>(base + SRC1[i]);
    };
    auto load_w2 = [&](int jc) {
        const uint8_t* base = p.w2 + (int64_t)jc * 32 * 2 + o2;
#pragma unroll
        for (int i = 0; i < N2; ++i) s2[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)i * 128 * G::HID * 2);
    };
    auto store_w1 = [&](uint8_t* st) {
#pragma unroll
        for (int i = 0; i < N1; ++i) *reinterpret_cast<u32x4*>(st + d1 + i * 16 * G::ROWB1) = s1[i];
    };
    auto store_w2 = [&](uint8_t* st) {
#pragma unroll
        for (int i = 0; i < N2; ++i) {
            const u32x2 lo = {s2[i][0], s2[i][1]}, hi = {s2[i][2], s2[i][3]};
            *reinterpret_cast<u32x2*>(st + d2 + i * 128 * G::ROWB2) = lo;
            *reinterpret_cast<u32x2*>(st + d2 + i * 128 * G::ROWB2 + 8) = hi;
        }
    };
    load_w1(0);
    load_w2(0);
    for (int i = tid; i < 2 * G::HID; i += NTH) lb1[i] = p.b1 ? ld_elem<DT>(p.b1, i) : 0.f;
    for (int i = tid; i < G::C; i += NTH) lb2[i] = p.b2 ? ld_elem<DT>(p.b2, i) : 0.f;
    // the "chunk -1" the first iteration's second GEMM consumes: zero activations against a zeroed weight slot
    for (int i = tid; i < G::HX_BYTES / 4; i += NTH) reinterpret_cast<uint32_t*>(hxs + G::HX_BYTES)[i] = 0u;
    for (int i = tid; i < G::W2_BYTES / 4; i += NTH) reinterpret_cast<uint32_t*>(w2s + 2 * G::W2_BYTES)[i] = 0u;

    typename E::v8 xf[KC];
    load_panel<DT, KC>(xf, p.x, G::C, p.M, mw0, l31, half);
    if (LN) layernorm_panel<DT, KC>(xf, p.gamma, p.beta, p.eps, l31, half);

    f32x16 yacc[G::CT];
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ct][r] = 0.f;

    store_w1(w1s);
    store_w2(w2s);
    load_w1(1);
    store_w1(w1s + G::W1_BYTES);
    __syncthreads();

    const int wrow = (32 * hh + l31) * G::ROWB1 + half * 16;  // this wave's rows of a W1 stage
    // (b1 as the accumulators' initial value: see mlp_kernel)
    auto bias_init = [&](int jc, f32x16& a) {
        const int u0 = jc * 32 + 16 * hh + 4 * half;
        const float4 v0 = *reinterpret_cast<const float4*>(lb1 + u0), v1 = *reinterpret_cast<const float4*>(lb1 + u0 + 8);
        const float4 g0 = *reinterpret_cast<const float4*>(lb1 + G::HID + u0), g1 = *reinterpret_cast<const float4*>(lb1 + G::HID + u0 + 8);
        a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w; a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
        a[8] = g0.x; a[9] = g0.y; a[10] = g0.z; a[11] = g0.w; a[12] = g1.x; a[13] = g1.y; a[14] = g1.z; a[15] = g1.w;
    };
    f32x16 acur;
    {
        bias_init(0, acur);
        typename E::v8 wf[4][1];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            rp_load_group<DT, KC>(wf, w1s + wrow, g * 4);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) acur = E::mfma32(wf[cc][0], xf[g * 4 + cc], acur);
        }
    }
    __syncthreads();

    const int w2row = (128 * hh + l31) * G::ROWB2 + half * 8;
    const int hxoff = pair * 2048 + lane * 16;
    auto gemm2 = [&](int slot, int parity) {  // y (this wave's 128 columns) += W2[:, chunk] . h(chunk), both halves of the chunk
        const uint8_t* w2t = w2s + slot * G::W2_BYTES + w2row;
        const uint8_t* hx = hxs + parity * G::HX_BYTES + hxoff;
        const typename E::v8 h0 = as_v8<DT>(*reinterpret_cast<const uint4*>(hx));
        const typename E::v8 h1 = as_v8<DT>(*reinterpret_cast<const uint4*>(hx + 1024));
#pragma unroll
        for (int ct = 0; ct < G::CT; ++ct) {
            const uint8_t* wp = w2t + ct * 32 * G::ROWB2;
            const uint2 a0 = *reinterpret_cast<const uint2*>(wp), a1 = *reinterpret_cast<const uint2*>(wp + 16);
            const uint2 c0 = *reinterpret_cast<const uint2*>(wp + 32), c1 = *reinterpret_cast<const uint2*>(wp + 48);
            yacc[ct] = E::mfma32(as_v8<DT>(make_uint4(a0.x, a0.y, a1.x, a1.y)), h0, yacc[ct]);
            yacc[ct] = E::mfma32(as_v8<DT>(make_uint4(c0.x, c0.y, c1.x, c1.y)), h1, yacc[ct]);
        }
    };

#ifndef MLP2_ABL
#define MLP2_ABL 0  // timing ablations (results are wrong): 1 = no global weight loads, 2 = no LDS weight stores, 4 = no per-chunk barrier, 8 = no GELU
#endif
    for (int jc = 0; jc < G::NCHUNK; ++jc) {
        if (!(MLP2_ABL & 1)) {
            load_w1(jc + 2 < G::NCHUNK ? jc + 2 : G::NCHUNK - 1);
            load_w2(jc + 1 < G::NCHUNK ? jc + 1 : G::NCHUNK - 1);
        }
        // operands of the second GEMM of chunk jc-1 (weights in slot (jc+2)%3, activations of both halves handed over at the last
        // barrier) and the first fragment group of the first GEMM of chunk jc+1: one batch of LDS reads
        typename E::v8 w2f[G::CT][2], hp[2];
        {
            const uint8_t* w2t = w2s + ((jc + 2) % 3) * G::W2_BYTES + w2row;
            const uint8_t* hx = hxs + ((jc + 1) & 1) * G::HX_BYTES + hxoff;
            hp[0] = as_v8<DT>(*reinterpret_cast<const uint4*>(hx));
            hp[1] = as_v8<DT>(*reinterpret_cast<const uint4*>(hx + 1024));
#pragma unroll
            for (int ct = 0; ct < G::CT; ++ct)
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    const uint8_t* wp = w2t + ct * 32 * G::ROWB2 + sh * 32;
                    const uint2 a0 = *reinterpret_cast<const uint2*>(wp), a1 = *reinterpret_cast<const uint2*>(wp + 16);
                    w2f[ct][sh] = as_v8<DT>(make_uint4(a0.x, a0.y, a1.x, a1.y));
                }
        }
        f32x16 anxt;
        bias_init(jc + 1 < G::NCHUNK ? jc + 1 : jc, anxt);
        typename E::v8 hb;
        auto geglu_step = [&](int r) {
            const float v0 = acur[r];
            const float v1 = acur[r + 1];
            const apad_f32x2 gt = {acur[8 + r], acur[9 + r]};
            const apad_f32x2 ge = (MLP2_ABL & 8) ? gt : gelu_erf_2(gt);
            float pr0 = v0 * ge[0], pr1 = v1 * ge[1];  // (fp32 product, then one rounding: see mlp_kernel)
            asm volatile("" : "+v"(pr0), "+v"(pr1));
            hb[r] = (typename E::elem)pr0;
            hb[r + 1] = (typename E::elem)pr1;
        };
        const uint8_t* wt = w1s + ((jc + 1) & 1) * G::W1_BYTES + wrow;
        typename E::v8 wfa[4][1], wfb[4][1];
        rp_load_group<DT, KC>(wfa, wt, 0);
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            rp_load_group<DT, KC>(wfb, wt, (g + 1) * 4);
            yacc[g] = E::mfma32(w2f[g][0], hp[0], yacc[g]);
            yacc[g] = E::mfma32(w2f[g][1], hp[1], yacc[g]);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) anxt = E::mfma32(wfa[cc][0], xf[g * 4 + cc], anxt);
            geglu_step(2 * g);
            if (g + 2 < NG) rp_load_group<DT, KC>(wfa, wt, (g + 2) * 4);
            yacc[g + 1] = E::mfma32(w2f[g + 1][0], hp[0], yacc[g + 1]);
            yacc[g + 1] = E::mfma32(w2f[g + 1][1], hp[1], yacc[g + 1]);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) anxt = E::mfma32(wfb[cc][0], xf[(g + 1) * 4 + cc], anxt);
            geglu_step(2 * g + 2);
        }
#ifndef MLP2_VALU
#define MLP2_VALU 0
#endif
#if MLP2_VALU > 0
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, MLP2_VALU, 0);
        }
#endif
        *reinterpret_cast<uint4*>(hxs + (jc & 1) * G::HX_BYTES + hxoff + hh * 1024) = as_u4<DT>(hb);
        if (!(MLP2_ABL & 2)) {
            store_w1(w1s + (jc & 1) * G::W1_BYTES);
            store_w2(w2s + ((jc + 1) % 3) * G::W2_BYTES);
        }
        if (!(MLP2_ABL & 4)) __syncthreads();
        acur = anxt;
    }
    gemm2((G::NCHUNK - 1) % 3, (G::NCHUNK - 1) & 1);
    __syncthreads();  // every wave is done with the weight stages: the W1 ring becomes the per-wave output scratch

    uint8_t* const scr = smem + wave * SCR_BYTES;
#pragma unroll
    for (int ct = 0; ct < G::CT; ++ct) {
        const int col0 = 128 * hh + ct * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(lb2 + col0 + 8 * g + 4 * half);
            typename E::v4 y;
            y[0] = (typename E::elem)(yacc[ct][4 * g + 0] + b4.x);
            y[1] = (typename E::elem)(yacc[ct][4 * g + 1] + b4.y);
            y[2] = (typename E::elem)(yacc[ct][4 * g + 2] + b4.z);
            y[3] = (typename E::elem)(yacc[ct][4 * g + 3] + b4.w);
            *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
        }
        scratch_flush<DT>(scr, 32, p.out, G::C, col0, p.x, G::C, mw0, p.M, lane);
    }
}

template <int DT, int KC, bool LN> int mlp2_launch(const MlpP& p, hipStream_t s) {
    using G = Mlp2Cfg<KC>;
    auto kern = mlp2_kernel<DT, KC, LN>;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), G::LDS, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.M + 127) / 128)), dim3(512), G::LDS, s, p);
    return apad_check_launch("apad_geglu_mlp");
}

template <int DT, int KC> int mlp_dispatch(const MlpP& p, bool ln, hipStream_t s) {
    // one form for every launch this entry point serves (round 5): the one-wave-per-SIMD 64- / 128-token kernel of rounds 1-4 measured slower at
    // every size (4 000 rows: 80.6 vs 65.7 us; 16 000: 86.4 vs 67.1) and was removed; launches of >= 48 000 rows take apad_geglu_mlp_packed (mlp3.hip)
    return ln ? mlp2_launch<DT, KC, true>(p, s) : mlp2_launch<DT, KC, false>(p, s);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int apad_geglu_mlp(const apad_mlp_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_geglu_mlp: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_geglu_mlp: dtype %d not supported", d->dtype);
    APAD_CHECK(d->x && d->w1 && d->w2 && d->out && d->M > 0, "apad_geglu_mlp: null operand / empty problem");
    APAD_CHECK(al16(d->x) && al16(d->w1) && al16(d->w2) && al16(d->out) && al16(d->ln_gamma) && al16(d->ln_beta),
               "apad_geglu_mlp: pointers must be 16-byte aligned");
    if (d->C != 256) {
        apad_set_error("apad_geglu_mlp: C=%d outside the kernel envelope (256)", d->C);
        return -3;
    }
    const bool ln = d->ln_gamma != nullptr;
    if (ln) APAD_CHECK(d->ln_beta != nullptr, "apad_geglu_mlp: LayerNorm needs gamma and beta");
    MlpP p;
    p.x = (const uint8_t*)d->x; p.gamma = (const uint8_t*)d->ln_gamma; p.beta = (const uint8_t*)d->ln_beta;
    p.w1 = (const uint8_t*)d->w1; p.b1 = (const uint8_t*)d->b1; p.w2 = (const uint8_t*)d->w2; p.b2 = (const uint8_t*)d->b2;
    p.out = (uint8_t*)d->out; p.M = d->M; p.eps = d->ln_eps;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? mlp_dispatch<APAD_BF16, 16>(p, ln, s) : mlp_dispatch<APAD_F16, 16>(p, ln, s);
}
