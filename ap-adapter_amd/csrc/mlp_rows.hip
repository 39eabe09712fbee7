// apad_geglu_mlp_rows: the fused feed-forward of a BasicTransformerBlock (apad_geglu_mlp's arithmetic)
//     out = x + W2 . ( value * gelu(gate) ) + b2,   [value | gate] = W1 . LayerNorm(x) + b1        (diffusers FeedForward / GEGLU)
// for the 384-wide level, whose x panel + output accumulators do not fit mlp2_kernel's registers.  Row-tile form (as
// attention.hip's xattn_rows_kernel): one 512-thread workgroup owns 64 tokens; the normalised tokens live in LDS for the whole pass, the
// hidden activation exists 128 units at a time in a double-buffered LDS chunk, the 64 x 384 output accumulates in registers (48 per
// lane), and BOTH weight matrices stream from L2 straight into registers as fragment-packed MFMA operands (one contiguous KB per
// wave-load, request streams that run two to three k-steps ahead and carry over from chunk to chunk).  Per 128-unit chunk:
//   A. wave (q = w & 3, panel = w >> 2): value tile and gate tile of hidden units 128 j + 32 q .. + 31 for its 32-token panel
//      (K = 384: 48 MFMAs) -> value * gelu(gate) in registers (same lanes / registers) -> bf16 -> the chunk buffer
//   (one workgroup barrier)
//   B. y^T[features 96 q .. + 95][panel] += W2[:, chunk] . h^T (K = 128: 24 MFMAs)
// The 8C-wide projection and the 4C-wide activation never reach HBM (the chain: 2 x 49.5 MB per launch at 16128 rows, two launches).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int MR_TM = 64, MR_HC = 128;
struct MrP {
    const uint8_t* x;
    const uint8_t* gamma;
    const uint8_t* beta;
    const uint8_t* w1;  // packed [8C / 32 row tiles][C / 16 k-steps][64 lanes][8]
    const uint8_t* b1;
    const uint8_t* w2;  // packed [C / 32 row tiles][4C / 16 k-steps][64 lanes][8]
    const uint8_t* b2;
    uint8_t* out;
    int64_t M;
    float eps;
};

template <int DT> __device__ __forceinline__ typename ET<DT>::v8 mr_ld(const uint8_t* p) { return as_v8<DT>(*reinterpret_cast<const uint4*>(p)); }

template <int DT, int C>
__global__ __launch_bounds__(512) void mlp_rows_kernel(MrP p) {
    using E = ET<DT>;
    constexpr int HID = 4 * C, NCH = HID / MR_HC, KS1 = C / 16, KS2 = HID / 16, KSC = MR_HC / 16, NT2 = C / 32 / 4;
    constexpr int ROWB = C * 2 + 16, HROWB = MR_HC * 2 + 16, HBYTES = MR_TM * HROWB, CH = C / 64;
    static_assert(C % 128 == 0 && KS1 % 3 == 0 && KSC % 2 == 0, "4 feature quarters of whole row tiles; request streams of 3 / 2 register sets");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const X = smem;
    uint8_t* const Hs = smem + MR_TM * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int wq = wave & 3, pnl = wave >> 2;
    const int64_t row0 = (int64_t)blockIdx.x * MR_TM;
    const int nrows = p.M - row0 < MR_TM ? (int)(p.M - row0) : MR_TM;
    const uint8_t* const xb = p.x + row0 * C * 2;

    // the two weight request streams start before the tokens arrive
    // W1: chunk j, k-step kk -> value tile 4 j + wq, gate tile HID / 32 + 4 j + wq
    auto w1v = [&](int j, int kk) { return p.w1 + ((int64_t)(4 * j + wq) * KS1 + kk) * 1024 + lane * 16; };
    auto w1g = [&](int j, int kk) { return p.w1 + ((int64_t)(HID / 32 + 4 * j + wq) * KS1 + kk) * 1024 + lane * 16; };
    auto w2a = [&](int n, int j, int kk) { return p.w2 + ((int64_t)(wq * NT2 + n) * KS2 + j * KSC + kk) * 1024 + lane * 16; };
    typename E::v8 fv[3], fg[3], f2[2][NT2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fv[i] = mr_ld<DT>(w1v(0, i));
        fg[i] = mr_ld<DT>(w1g(0, i));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int n = 0; n < NT2; ++n) f2[i][n] = mr_ld<DT>(w2a(n, 0, i));

    // ---- LayerNorm -> X (8 lanes per row, two passes in registers) ----
    {
        const int sub = tid & 7, row = tid >> 3;
        float v[CH][8];
        const bool ok = row < nrows;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            uint4 u = make_uint4(0u, 0u, 0u, 0u);
            if (ok) u = *reinterpret_cast<const uint4*>(xb + ((int64_t)row * C + (sub + 8 * i) * 8) * 2);
            unpack8<DT>(u, v[i]);
        }
        if (p.gamma != nullptr) {
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) s1 += v[i][e];
            s1 += __shfl_xor(s1, 1);
            s1 += __shfl_xor(s1, 2);
            s1 += __shfl_xor(s1, 4);
            const float mean = s1 * (1.0f / C);
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < CH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dd = v[i][e] - mean;
                    s2 = __builtin_fmaf(dd, dd, s2);
                }
            s2 += __shfl_xor(s2, 1);
            s2 += __shfl_xor(s2, 2);
            s2 += __shfl_xor(s2, 4);
            const float rstd = rsqrtf(s2 * (1.0f / C) + p.eps);
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                float g[8], be[8];
                unpack8<DT>(*reinterpret_cast<const uint4*>(p.gamma + (sub + 8 * i) * 16), g);
                unpack8<DT>(*reinterpret_cast<const uint4*>(p.beta + (sub + 8 * i) * 16), be);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[i][e] = ok ? (v[i][e] - mean) * rstd * g[e] + be[e] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) *reinterpret_cast<uint4*>(X + row * ROWB + (sub + 8 * i) * 16) = pack8<DT>(v[i]);
    }
    __syncthreads();

    f32x16 yacc[NT2];
#pragma unroll
    for (int n = 0; n < NT2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[n][r] = 0.f;
    const uint8_t* const xs = X + (pnl * 32 + l31) * ROWB + half * 16;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int j = 0; j < NCH; ++j) {
        uint8_t* const Hb = Hs + (j & 1) * HBYTES;
        // ---- A. value / gate tiles of this wave's 32 hidden units x 32 tokens ----
        f32x16 av = zero16, ag = zero16;
#pragma unroll 1
        for (int kk = 0; kk < KS1; kk += 3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const typename E::v8 t = mr_ld<DT>(xs + (kk + i) * 32);
                av = E::mfma32(fv[i], t, av);
                ag = E::mfma32(fg[i], t, ag);
                // the stream carries over into the next chunk's first k-steps
                const int nk = kk + i + 3;
                const int jn = nk < KS1 ? j : j + 1, kn = nk < KS1 ? nk : nk - KS1;
                if (jn < NCH) {
                    fv[i] = mr_ld<DT>(w1v(jn, kn));
                    fg[i] = mr_ld<DT>(w1g(jn, kn));
                }
            }
        }
        {
            const int u0 = j * MR_HC + wq * 32;  // first hidden unit of this wave's tile
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int u = u0 + 8 * g + 4 * half;
                typename E::v4 h;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    float bv0 = 0.f, bv1 = 0.f, bg0 = 0.f, bg1 = 0.f;
                    if (p.b1 != nullptr) {
                        bv0 = ld_elem<DT>(p.b1, u + e);
                        bv1 = ld_elem<DT>(p.b1, u + e + 1);
                        bg0 = ld_elem<DT>(p.b1, HID + u + e);
                        bg1 = ld_elem<DT>(p.b1, HID + u + e + 1);
                    }
                    const apad_f32x2 ge = gelu_erf_2((apad_f32x2){ag[4 * g + e] + bg0, ag[4 * g + e + 1] + bg1});
                    h[e] = (typename E::elem)((av[4 * g + e] + bv0) * ge[0]);
                    h[e + 1] = (typename E::elem)((av[4 * g + e + 1] + bv1) * ge[1]);
                }
                *reinterpret_cast<uint2*>(Hb + (pnl * 32 + l31) * HROWB + (wq * 32 + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, h);
            }
        }
        __syncthreads();  // the chunk is complete; (a wave past this point has also finished phase B of chunk j - 1: the other buffer is free)
        // ---- B. y^T += W2[:, chunk] . h^T ----
        const uint8_t* const hs = Hb + (pnl * 32 + l31) * HROWB + half * 16;
#pragma unroll 1
        for (int kk = 0; kk < KSC; kk += 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const typename E::v8 t = mr_ld<DT>(hs + (kk + i) * 32);
#pragma unroll
                for (int n = 0; n < NT2; ++n) yacc[n] = E::mfma32(f2[i][n], t, yacc[n]);
                const int nk = kk + i + 2;
                const int jn = nk < KSC ? j : j + 1, kn = nk < KSC ? nk : nk - KSC;
                if (jn < NCH) {
#pragma unroll
                    for (int n = 0; n < NT2; ++n) f2[i][n] = mr_ld<DT>(w2a(n, jn, kn));
                }
            }
        }
    }
    __syncthreads();  // X is dead: it becomes the output tile

    // ---- y + b2 -> X (rounded like the chain's GEMM output), then + residual -> out ----
#pragma unroll
    for (int n = 0; n < NT2; ++n) {
        const int f0 = (wq * NT2 + n) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = f0 + 8 * g + 4 * half;
            typename E::v4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (typename E::elem)(yacc[n][4 * g + e] + (p.b2 ? ld_elem<DT>(p.b2, f + e) : 0.f));
            *reinterpret_cast<uint2*>(X + (pnl * 32 + l31) * ROWB + f * 2) = __builtin_bit_cast(uint2, y);
        }
    }
    __syncthreads();
    uint8_t* const ob = p.out + row0 * C * 2;
    constexpr int CPR = C / 8;
    for (int idx = tid; idx < MR_TM * CPR; idx += 512) {
        const int row = idx / CPR, ch = idx - row * CPR;
        if (row >= nrows) break;
        float y[8], r[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(X + row * ROWB + ch * 16), y);
        unpack8<DT>(*reinterpret_cast<const uint4*>(xb + ((int64_t)row * C + ch * 8) * 2), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] += r[e];
        *reinterpret_cast<uint4*>(ob + ((int64_t)row * C + ch * 8) * 2) = pack8<DT>(y);
    }
}

template <int DT, int C> int mlp_rows_launch(const MrP& p, hipStream_t s) {
    constexpr int LDS = MR_TM * (C * 2 + 16) + 2 * MR_TM * (MR_HC * 2 + 16);
    auto kern = mlp_rows_kernel<DT, C>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.M + MR_TM - 1) / MR_TM)), dim3(512), LDS, s, p);
    return apad_check_launch("apad_geglu_mlp_rows");
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int apad_geglu_mlp_rows(const apad_mlp_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_geglu_mlp_rows: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_geglu_mlp_rows: dtype %d not supported (16-bit only)", d->dtype);
    APAD_CHECK(d->x && d->w1 && d->w2 && d->out && d->M > 0, "apad_geglu_mlp_rows: null operand / empty problem");
    APAD_CHECK(al16(d->x) && al16(d->w1) && al16(d->w2) && al16(d->out) && al16(d->ln_gamma) && al16(d->ln_beta),
               "apad_geglu_mlp_rows: pointers must be 16-byte aligned");
    if (d->C != 384) {
        apad_set_error("apad_geglu_mlp_rows: C=%d outside the kernel envelope (384)", d->C);
        return -3;
    }
    APAD_CHECK((d->ln_gamma == nullptr) == (d->ln_beta == nullptr), "apad_geglu_mlp_rows: LayerNorm needs gamma and beta");
    MrP p;
    p.x = (const uint8_t*)d->x; p.gamma = (const uint8_t*)d->ln_gamma; p.beta = (const uint8_t*)d->ln_beta;
    p.w1 = (const uint8_t*)d->w1; p.b1 = (const uint8_t*)d->b1; p.w2 = (const uint8_t*)d->w2; p.b2 = (const uint8_t*)d->b2;
    p.out = (uint8_t*)d->out; p.M = d->M; p.eps = d->ln_eps;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? mlp_rows_launch<APAD_BF16, 384>(p, s) : mlp_rows_launch<APAD_F16, 384>(p, s);
}
