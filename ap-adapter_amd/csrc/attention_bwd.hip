// apad_attention_bwd: gradients of one softmax-attention segment, O = softmax(Q K^T * scale + bias) V, for the
// training step of the adapter (reference: train_apadapter_v2.py:941-957 backpropagates through every
// F.scaled_dot_product_attention of attention_processor.py:250-253 / :429-445).
//
// Flash-style recomputation: P is rebuilt from Q, K and the log-sum-exp the forward kernel stored, so nothing of
// size N x L is ever saved.  Two passes share one shape -- "a lane owns one row of the result, the other sequence is
// walked in 32-wide tiles, the score-type MFMA output (C layout) is reused in registers as the B operand of the
// accumulate-type MFMA by reading the other operand with the permuted k order" -- exactly the forward kernel's trick:
//   dQ pass   (lane = query):  S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T o (dP^T - delta),  dQ^T += K^T dS^T
//   dKV pass  (lane = key):    S = Q K^T,    dP = dO V^T,    dS = P o (dP - delta),        dV^T += dO^T P,  dK^T += Q^T dS
// The transposed operands (K^T for the dQ pass; Q^T and dO^T for the dKV pass) are [B][H][D][pad] copies made by
// apad_head_transpose, so every fragment is a 16- or 8-byte vector load.  Fragments are loaded straight from global
// memory (L2-resident at training batch sizes); there is no LDS staging and no barrier.
#include "common.h"
#include "f32_ops.h"

namespace {

constexpr float LOG2E_B = 1.4426950408889634f;

struct BwdP {
    const uint8_t* q;
    const uint8_t* k;
    const uint8_t* v;
    const uint8_t* qt;
    const uint8_t* kt;
    const uint8_t* out;
    const uint8_t* dout;
    const uint8_t* doutt;
    const float* lse;
    const float* key_bias;
    float* delta;
    uint8_t* dq;
    uint8_t* dk;
    uint8_t* dv;
    int32_t B, N, H, L, Npad, Lpad;
    float scale, scale_log2, dout_scale;
    int32_t accumulate_dq;
    int32_t ldg;  // row stride (elements) of dq / dk / dv: H * D, or 3 H D when the three are column blocks of one [rows][3C] buffer
};

// A-operand fragment (row = tile row l31, k = 8 contiguous elements) of a row-major [rows][C] matrix; rows >= limit read 0
template <int DT> __device__ __forceinline__ typename ET<DT>::v8 row_frag(const uint8_t* base, int64_t row, int limit, int C, int col) {
    uint4 u = make_uint4(0u, 0u, 0u, 0u);
    if (row < limit) u = *reinterpret_cast<const uint4*>(base + (row * C + col) * 2);
    return as_v8<DT>(u);
}

// A-operand fragment with the C-layout k permutation from a transposed [D][pad] matrix: row = d (zero for d >= D),
// k slots 0..3 <-> columns c0 + 4*half + (0..3), slots 4..7 <-> c0 + 8 + 4*half + (0..3)
template <int DT, int D> __device__ __forceinline__ typename ET<DT>::v8 tr_frag(const uint8_t* base, int d, int pad, int c0, int half) {
    uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
    if (d < D) {
        const uint8_t* ptr = base + ((int64_t)d * pad + c0 + 4 * half) * 2;
        lo = *reinterpret_cast<const uint2*>(ptr);
        hi = *reinterpret_cast<const uint2*>(ptr + 16);
    }
    return as_v8<DT>(make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// write one lane-owned row of a transposed accumulator set: acc[t] rows = d (32t + (r&3) + 8(r>>2) + 4half), col = lane's row
template <int DT, int D>
__device__ __forceinline__ void store_row(uint8_t* dst /* row base + head offset */, const f32x16* acc, float mul, int half, bool accumulate) {
    using E = ET<DT>;
    constexpr int TT = (D + 31) / 32;
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * t + 8 * g + 4 * half;
            if (d0 < D) {
                float vals[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) vals[j] = acc[t][4 * g + j] * mul;
                uint2* ptr = reinterpret_cast<uint2*>(dst + d0 * 2);
                if (accumulate) {
                    typename E::v4 old = __builtin_bit_cast(typename E::v4, *ptr);
#pragma unroll
                    for (int j = 0; j < 4; ++j) vals[j] += (float)old[j];
                }
                typename E::v4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)vals[j];
                *ptr = __builtin_bit_cast(uint2, pk);
            }
        }
}

// Branch-free forms for the software-pipelined loops below: out-of-range rows are CLAMPED (their scores are masked to probability 0
// where they matter, and accumulator rows d >= D are never stored), so every load is unconditional -- an exec-masked load makes the
// compiler's wait-count pass wait for everything in flight, which would undo the prefetch.
template <int DT> __device__ __forceinline__ typename ET<DT>::v8 row_frag_c(const uint8_t* base, int row, int limit, int C, int col) {
    const int r = row < limit ? row : limit - 1;
    return as_v8<DT>(*reinterpret_cast<const uint4*>(base + ((int64_t)r * C + col) * 2));
}
template <int DT, int D> __device__ __forceinline__ typename ET<DT>::v8 tr_frag_c(const uint8_t* base, int d, int pad, int c0, int half) {
    const int dd = d < D ? d : D - 1;
    const uint8_t* ptr = base + ((int64_t)dd * pad + c0 + 4 * half) * 2;
    const uint2 lo = *reinterpret_cast<const uint2*>(ptr), hi = *reinterpret_cast<const uint2*>(ptr + 16);
    return as_v8<DT>(make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// ---------------------------------------------------------------------------------------------------------------------
template <int DT, int D> __global__ __launch_bounds__(256) void dq_kernel(BwdP p) {
    using E = ET<DT>;
    constexpr int KC = D / 16, TT = (D + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int C = p.H * D;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (q0 >= p.N) return;
    const int qi = q0 + l31;
    const bool qvalid = qi < p.N;
    const int64_t qrow = (int64_t)b * p.N + (qvalid ? qi : p.N - 1);

    // resident B operands: Q^T and dO^T fragments of the lane's query (k = 8 contiguous head-dim elements)
    typename E::v8 qf[KC], gf[KC];
#pragma unroll
    for (int s = 0; s < KC; ++s) {
        qf[s] = as_v8<DT>(*reinterpret_cast<const uint4*>(p.q + (qrow * C + h * D + 16 * s + 8 * half) * 2));
        gf[s] = as_v8<DT>(*reinterpret_cast<const uint4*>(p.dout + (qrow * C + h * D + 16 * s + 8 * half) * 2));
    }
    const int64_t stat = ((int64_t)b * p.H + h) * p.Npad + (qvalid ? qi : 0);
    const float lse2 = p.lse[stat];
    // delta[q] = dout_scale * sum_d dO[q][d] O[q][d], formed here from the resident dO fragments (a lane holds half of its query's
    // head dims: one half-wave exchange) and written for the dK/dV pass that follows on the stream -- pad entries 0 -- instead of by a
    // launch of its own (the step at batch 4 is bound by its launch count: 256 launches)
    float delta;
    {
        float dsum = 0.f;
#pragma unroll
        for (int s = 0; s < KC; ++s) {
            const typename E::v8 of = as_v8<DT>(*reinterpret_cast<const uint4*>(p.out + (qrow * C + h * D + 16 * s + 8 * half) * 2));
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum += (float)of[j] * (float)gf[s][j];
        }
        delta = half_sum(dsum) * p.dout_scale;
        if (half == 0 && qi < p.Npad) p.delta[((int64_t)b * p.H + h) * p.Npad + qi] = qvalid ? delta : 0.f;
    }
    const uint8_t* kb = p.k + ((int64_t)b * p.L * C + h * D) * 2;
    const uint8_t* vb = p.v + ((int64_t)b * p.L * C + h * D) * 2;
    const uint8_t* ktb = p.kt + ((int64_t)b * p.H + h) * D * p.Lpad * 2;
    const float* bias = p.key_bias ? p.key_bias + (int64_t)b * p.L : nullptr;

    f32x16 acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // key loop, software-pipelined by hand (round 3): the K / V / K^T fragments of tile t + 1 are requested before tile t is
    // computed (two register sets, the loop unrolled by two).  The first version loaded each fragment right in front of its MFMA:
    // at the training batch of 4 the pass is a chain of exposed L2 latencies, not MFMA work.
    struct Fr {
        typename E::v8 kf[KC], vf[KC], ktf[TT][2];
    };
    const int last0 = ((p.L - 1) / 32) * 32;  // first key of the last tile
    auto load = [&](Fr& f, int k0) {
#pragma unroll
        for (int s_ = 0; s_ < KC; ++s_) {
            f.kf[s_] = row_frag_c<DT>(kb, k0 + l31, p.L, C, 16 * s_ + 8 * half);
            f.vf[s_] = row_frag_c<DT>(vb, k0 + l31, p.L, C, 16 * s_ + 8 * half);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) f.ktf[t][u] = tr_frag_c<DT, D>(ktb, 32 * t + l31, p.Lpad, k0 + 16 * u, half);
    };
    auto compute = [&](const Fr& f, int k0) {
        f32x16 st, dpt;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < KC; ++s_) {
            st = E::mfma32(f.kf[s_], qf[s_], st);
            dpt = E::mfma32(f.vf[s_], gf[s_], dpt);
        }
        typename E::v8 dsf[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float e = st[r] * p.scale_log2 - lse2;
            if (bias) e += bias[key < p.L ? key : p.L - 1] * LOG2E_B;
            const float pr = (key < p.L && qvalid) ? __builtin_amdgcn_exp2f(e) : 0.f;
            dsf[r >> 3][r & 7] = (typename E::elem)(pr * (dpt[r] * p.dout_scale - delta));
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t] = E::mfma32(f.ktf[t][u], dsf[u], acc[t]);
    };
    Fr fa, fb;
    load(fa, 0);
    for (int k0 = 0; k0 < p.L; k0 += 64) {
        load(fb, k0 + 32 <= last0 ? k0 + 32 : last0);  // (unconditional: the last tile is re-requested)
        compute(fa, k0);
        if (k0 + 32 >= p.L) break;
        load(fa, k0 + 64 <= last0 ? k0 + 64 : last0);
        compute(fb, k0 + 32);
    }
    if (qvalid) store_row<DT, D>(p.dq + (qrow * p.ldg + h * D) * 2, acc, p.scale, half, p.accumulate_dq != 0);
}

// ---------------------------------------------------------------------------------------------------------------------
template <int DT, int D> __global__ __launch_bounds__(256) void dkv_kernel(BwdP p) {
    using E = ET<DT>;
    constexpr int KC = D / 16, TT = (D + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, b = bh / p.H, h = bh % p.H;
    const int C = p.H * D;
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    if (k0 >= p.L) return;
    const int ki = k0 + l31;
    const bool kvalid = ki < p.L;
    const int64_t krow = (int64_t)b * p.L + (kvalid ? ki : p.L - 1);

    typename E::v8 kf[KC], vf[KC];
#pragma unroll
    for (int s = 0; s < KC; ++s) {
        kf[s] = as_v8<DT>(*reinterpret_cast<const uint4*>(p.k + (krow * C + h * D + 16 * s + 8 * half) * 2));
        vf[s] = as_v8<DT>(*reinterpret_cast<const uint4*>(p.v + (krow * C + h * D + 16 * s + 8 * half) * 2));
    }
    const float kbias = p.key_bias ? p.key_bias[(int64_t)b * p.L + (kvalid ? ki : p.L - 1)] * LOG2E_B : 0.f;
    const uint8_t* qb = p.q + ((int64_t)b * p.N * C + h * D) * 2;
    const uint8_t* gb = p.dout + ((int64_t)b * p.N * C + h * D) * 2;
    const uint8_t* qtb = p.qt + ((int64_t)b * p.H + h) * D * p.Npad * 2;
    const uint8_t* gtb = p.doutt + ((int64_t)b * p.H + h) * D * p.Npad * 2;
    const float* lse = p.lse + ((int64_t)b * p.H + h) * p.Npad;
    const float* del = p.delta + ((int64_t)b * p.H + h) * p.Npad;

    f32x16 dk[TT], dv[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[t][r] = dv[t][r] = 0.f;

    // query loop, software-pipelined like the dQ pass: Q / dO / Q^T / dO^T fragments and the row statistics of tile t + 1 are
    // requested before tile t is computed
    struct Fr {
        typename E::v8 qa[KC], ga[KC], gt[TT][2], qt[TT][2];
        float4 l4[4], d4[4];
    };
    const int last0 = ((p.N - 1) / 32) * 32;
    auto load = [&](Fr& f, int q0) {
#pragma unroll
        for (int s_ = 0; s_ < KC; ++s_) {
            f.qa[s_] = row_frag_c<DT>(qb, q0 + l31, p.N, C, 16 * s_ + 8 * half);
            f.ga[s_] = row_frag_c<DT>(gb, q0 + l31, p.N, C, 16 * s_ + 8 * half);
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f.gt[t][u] = tr_frag_c<DT, D>(gtb, 32 * t + l31, p.Npad, q0 + 16 * u, half);
                f.qt[t][u] = tr_frag_c<DT, D>(qtb, 32 * t + l31, p.Npad, q0 + 16 * u, half);
            }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // queries q0 + 8g + 4half + (0..3): the stats arrays are padded to Npad (multiple of 32), so the float4 is in bounds
            f.l4[g] = *reinterpret_cast<const float4*>(lse + q0 + 8 * g + 4 * half);
            f.d4[g] = *reinterpret_cast<const float4*>(del + q0 + 8 * g + 4 * half);
        }
    };
    auto compute = [&](const Fr& f, int q0) {
        f32x16 sc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = dp[r] = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < KC; ++s_) {
            sc = E::mfma32(f.qa[s_], kf[s_], sc);
            dp = E::mfma32(f.ga[s_], vf[s_], dp);
        }
        typename E::v8 pf[2], dsf[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float lv[4] = {f.l4[g].x, f.l4[g].y, f.l4[g].z, f.l4[g].w}, dl[4] = {f.d4[g].x, f.d4[g].y, f.d4[g].z, f.d4[g].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 4 * g + j, qq = q0 + 8 * g + 4 * half + j;
                const float e = sc[r] * p.scale_log2 + kbias - lv[j];
                const bool ok = qq < p.N && kvalid;  // the padded tail of the stats arrays is never trusted
                const float pr = ok ? __builtin_amdgcn_exp2f(e) : 0.f;
                pf[r >> 3][r & 7] = (typename E::elem)pr;
                dsf[r >> 3][r & 7] = (typename E::elem)(ok ? pr * (dp[r] * p.dout_scale - dl[j]) : 0.f);
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dv[t] = E::mfma32(f.gt[t][u], pf[u], dv[t]);
                dk[t] = E::mfma32(f.qt[t][u], dsf[u], dk[t]);
            }
    };
    Fr fa, fb;
    load(fa, 0);
    for (int q0 = 0; q0 < p.N; q0 += 64) {
        load(fb, q0 + 32 <= last0 ? q0 + 32 : last0);
        compute(fa, q0);
        if (q0 + 32 >= p.N) break;
        load(fa, q0 + 64 <= last0 ? q0 + 64 : last0);
        compute(fb, q0 + 32);
    }
    if (kvalid) {
        store_row<DT, D>(p.dk + (krow * p.ldg + h * D) * 2, dk, p.scale, half, false);
        store_row<DT, D>(p.dv + (krow * p.ldg + h * D) * 2, dv, p.dout_scale, half, false);
    }
}

template <int DT, int D> int launch_bwd(const BwdP& p, hipStream_t s) {
    hipLaunchKernelGGL((dq_kernel<DT, D>), dim3((unsigned)((p.N + 127) / 128), (unsigned)(p.B * p.H)), dim3(256), 0, s, p);
    if (p.dk != nullptr)
        hipLaunchKernelGGL((dkv_kernel<DT, D>), dim3((unsigned)((p.L + 127) / 128), (unsigned)(p.B * p.H)), dim3(256), 0, s, p);
    return apad_check_launch("apad_attention_bwd");
}

template <int DT> int launch_bwd_dt(const BwdP& p, int D, hipStream_t s) {
    switch (D) {
        case 16: return launch_bwd<DT, 16>(p, s);
        case 32: return launch_bwd<DT, 32>(p, s);
        case 48: return launch_bwd<DT, 48>(p, s);
        case 64: return launch_bwd<DT, 64>(p, s);
        case 80: return launch_bwd<DT, 80>(p, s);
    }
    apad_set_error("apad_attention_bwd: head dim %d not supported (16,32,48,64,80)", D);
    return -1;
}

// x [B][N][H*D] -> xt [B][H][D][pad], columns n >= N zero-filled: the transposed operand layout of both attention passes.
// Up to three tensors of one shape per launch (q, k and dO of a self-attention backward): blockIdx.z = tensor * B + b.
struct HtP {
    const uint8_t* x[3];
    uint8_t* xt[3];
};
template <int DT> __global__ __launch_bounds__(256) void head_transpose_kernel(HtP ptrs, int B, int N, int H, int D, int pad) {
    using E = ET<DT>;
    __shared__ typename E::elem tile[32][33];
    const int which = blockIdx.z / B;
    const uint8_t* x = which == 0 ? ptrs.x[0] : (which == 1 ? ptrs.x[1] : ptrs.x[2]);
    uint8_t* xt = which == 0 ? ptrs.xt[0] : (which == 1 ? ptrs.xt[1] : ptrs.xt[2]);
    const int b = blockIdx.z - which * B, c0 = blockIdx.y * 32, n0 = blockIdx.x * 32;  // c = h*D + d runs over H*D
    const int C = H * D;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const typename E::elem* xin = reinterpret_cast<const typename E::elem*>(x);
    typename E::elem* xo = reinterpret_cast<typename E::elem*>(xt);
    for (int i = ty; i < 32; i += 8) {
        const int n = n0 + i, c = c0 + tx;
        tile[i][tx] = (n < N && c < C) ? xin[((int64_t)b * N + n) * C + c] : (typename E::elem)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, n = n0 + tx;
        if (c < C && n < pad) xo[((int64_t)b * C + c) * pad + n] = tile[tx][i];
    }
}

}  // namespace

extern "C" int apad_head_transpose3(const void* x0, void* xt0, const void* x1, void* xt1, const void* x2, void* xt2, int32_t B,
                                    int32_t N, int32_t H, int32_t D, int32_t pad, int32_t dtype, void* stream) {
    APAD_CHECK(x0 && xt0 && B > 0 && N > 0 && H > 0 && D > 0, "apad_head_transpose: null operand / empty problem");
    APAD_CHECK((x1 == nullptr) == (xt1 == nullptr) && (x2 == nullptr) == (xt2 == nullptr) && (x1 != nullptr || x2 == nullptr),
               "apad_head_transpose3: tensors are given in order, each with its destination");
    APAD_CHECK(pad >= N && pad % 32 == 0, "apad_head_transpose: pad must be >= N and a multiple of 32");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_head_transpose: dtype %d not supported", dtype);
    const int nt = x2 ? 3 : (x1 ? 2 : 1);
    HtP ptrs{{(const uint8_t*)x0, (const uint8_t*)x1, (const uint8_t*)x2}, {(uint8_t*)xt0, (uint8_t*)xt1, (uint8_t*)xt2}};
    dim3 grid((unsigned)(pad / 32), (unsigned)((H * D + 31) / 32), (unsigned)(B * nt));
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_F32)
        hipLaunchKernelGGL(head_transpose_kernel<APAD_F32>, grid, dim3(256), 0, s, ptrs, B, N, H, D, pad);
    else if (dtype == APAD_BF16)
        hipLaunchKernelGGL(head_transpose_kernel<APAD_BF16>, grid, dim3(256), 0, s, ptrs, B, N, H, D, pad);
    else
        hipLaunchKernelGGL(head_transpose_kernel<APAD_F16>, grid, dim3(256), 0, s, ptrs, B, N, H, D, pad);
    return apad_check_launch("apad_head_transpose");
}

extern "C" int apad_head_transpose(const void* x, void* xt, int32_t B, int32_t N, int32_t H, int32_t D, int32_t pad,
                                   int32_t dtype, void* stream) {
    return apad_head_transpose3(x, xt, nullptr, nullptr, nullptr, nullptr, B, N, H, D, pad, dtype, stream);
}

extern "C" int apad_attention_bwd(const apad_attn_bwd_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_attention_bwd: null descriptor");
    if (d->dtype == APAD_F32) return apad_f32_attention_bwd(d, (hipStream_t)stream);  // fp32 training mode (f32_ops.hip): no transposed operands
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_attention_bwd: dtype %d not supported", d->dtype);
    APAD_CHECK(d->q && d->k && d->v && d->kt && d->out && d->dout && d->lse && d->delta && d->dq,
               "apad_attention_bwd: null operand");
    APAD_CHECK(d->B > 0 && d->N > 0 && d->H > 0 && d->L > 0, "apad_attention_bwd: empty problem");
    APAD_CHECK(d->Npad >= d->N && d->Npad % 32 == 0 && d->Lpad >= d->L && d->Lpad % 32 == 0,
               "apad_attention_bwd: Npad / Lpad must cover N / L and be multiples of 32");
    APAD_CHECK((d->dk == nullptr) == (d->dv == nullptr), "apad_attention_bwd: dk and dv are requested together");
    if (d->dk) APAD_CHECK(d->qt && d->doutt, "apad_attention_bwd: dk/dv need the transposed q and dout");
    BwdP p;
    p.q = (const uint8_t*)d->q; p.k = (const uint8_t*)d->k; p.v = (const uint8_t*)d->v; p.qt = (const uint8_t*)d->qt;
    p.kt = (const uint8_t*)d->kt; p.out = (const uint8_t*)d->out; p.dout = (const uint8_t*)d->dout;
    p.doutt = (const uint8_t*)d->doutt; p.lse = d->lse; p.key_bias = d->key_bias; p.delta = d->delta;
    p.dq = (uint8_t*)d->dq; p.dk = (uint8_t*)d->dk; p.dv = (uint8_t*)d->dv;
    p.B = d->B; p.N = d->N; p.H = d->H; p.L = d->L; p.Npad = d->Npad; p.Lpad = d->Lpad;
    p.scale = d->softmax_scale; p.scale_log2 = d->softmax_scale * LOG2E_B; p.dout_scale = d->dout_scale;
    p.accumulate_dq = d->accumulate_dq;
    p.ldg = d->ld_grad > 0 ? d->ld_grad : d->H * d->D;
    APAD_CHECK(p.ldg >= d->H * d->D && p.ldg % 8 == 0, "apad_attention_bwd: ld_grad must be >= H * D and a multiple of 8");
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? launch_bwd_dt<APAD_BF16>(p, d->D, s) : launch_bwd_dt<APAD_F16>(p, d->D, s);
}
