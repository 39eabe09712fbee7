// apad_hs_attention + apad_hs_out: the attention sub-layers of the 64-token level (C = 640, 8 heads of 80; <= 64 tokens per sample) as
// TWO launches per sub-layer instead of the LayerNorm-folded q|k|v GEMM -> attention -> to_out GEMM chain of 64x64-tile launches
// (reference: attention_processor.py:214-294 / :387-457 behind BasicTransformerBlock's norm1 / norm2, modeling_audioldm2.py:1047-1058).
//
// What bounds that level: a sample's 64 tokens are ONE MFMA row-panel pair, so every weight byte a CU pulls from L2 is worth 64 MACs --
// exactly the ridge of the CU's 64 B/clk vector-memory path against its 4 x 1024 FLOP/clk of MFMA -- and the chain's tiled launches
// (k-tile: load -> LDS write -> barrier -> read -> 16 MFMAs, one or two tiles in flight) sit on exposed latencies (rocprof round 3:
// 56 .. 69 % of their wave-cycles parked).  Here the work of a sub-layer is SLICED BY HEADS so that every CU streams a disjoint quarter
// of the weights exactly once, straight into registers, against all 64 tokens of one sample held in LDS:
//
//   hs_attn_kernel   workgroup = (sample b, head pair p), 8 waves, grid 4 B (256 workgroups for the CFG batch of 64):
//     1. LayerNorm(x[b]) -> X tile in LDS (8 lanes per row, whole cache lines per instruction); the first weight fragments are requested
//        BEFORE it, so the weight stream is already running
//     2. self-attention: [q | k | v](pair p)^T = W_p . X^T -- 15 row tiles of 32 features (5 q, 5 k, 5 v; the to_q rows carry log2(e)/sqrt(d))
//        x 2 token panels x 40 k-steps = 1200 MFMAs; wave w owns tiles w and w + 8 for BOTH panels, so each packed 1 KB weight fragment
//        is loaded once per workgroup (NSET k-steps ahead, SGPR base + lane offset, counted vmcnt); q, k -> row-major LDS tiles, v -> V^T
//        cross-attention: q(pair p) only (5 tiles); K / V^T are the hoisted per-site projections in HBM / L2
//     3. attention: wave (h, panel) for the pair's two heads -- short_seg.h's one-tile-per-segment softmax, both segments of the adapter
//        (text + scale * audio, each branch rounded before the blend) or a masked T5 segment; K / V^T fragments from LDS (self) or L2
//     4. O(pair p) [64][160] -> HBM (the only activation write: 20 KB per workgroup)
//   hs_out_kernel    workgroup = (sample b, output-column quarter c): out = x + (O . Wo^T + b_o), O tile [64][640] in LDS, the 160 weight
//     rows of the quarter streamed the same way; also emits the row statistics the folded LayerNorm of the next GEMM wants.
//
// Why two launches and not one: to_out contracts over ALL heads.  Keeping it inside the head-sliced launch means four fp32 partial
// [64][640] slabs per sample (164 KB written per workgroup, 656 KB read back by each of the next sub-layer's four workgroups: more
// bytes through the 64 B/clk path than the weights themselves); the launch boundary is the cheaper all-to-all (1.5 us, 20 KB per
// workgroup each way).  A row's result never depends on its batch: one workgroup = one sample, fixed summation order.
#include <stdlib.h>
#include <mutex>
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

#include "short_seg.h"

#ifndef HS_NSET
#define HS_NSET 8  // register sets of weight fragments = k-steps a fragment is requested ahead of its use (per wave: NSET x NT KB in flight)
#endif
#ifndef HS_ABL
#define HS_ABL 0  // timing ablations (tools/ab_build.sh; results are wrong): 1 = weight fragments loaded once, 2 = no attention phase, 4 = no projection MFMAs
#endif

// probe build (tools/ab_build.sh <tag> hsattn.hip -DHS_TRACE=<wave>; tools/hs_trace.py): s_memtime at the phase boundaries of one wave of every
// workgroup + the 100 MHz wall clock at its start / end.  Never part of the product library.
#ifdef HS_TRACE
__device__ unsigned long long hs_trace_buf[2][1024][16];
#define HS_STAMP(k_, i_)                                                                                   \
    if (lane == 0 && wave == (HS_TRACE) && blockIdx.x < 1024) {                                            \
        hs_trace_buf[k_][blockIdx.x][i_] = __builtin_amdgcn_s_memtime();                                   \
        if ((i_) == 0) hs_trace_buf[k_][blockIdx.x][14] = wall_clock64();                                  \
        if ((i_) == 9) hs_trace_buf[k_][blockIdx.x][15] = wall_clock64();                                  \
    }
#else
#define HS_STAMP(k_, i_)
#endif

// wave-uniform global pointer pinned to SGPRs (xattn.hip): loads take the scalar base + 32-bit lane offset form
typedef const __attribute__((address_space(1))) uint8_t* hs_gptr;
typedef const __attribute__((address_space(1))) u32x4* hs_gptr16;
__device__ __forceinline__ hs_gptr sgpr_ptr(const uint8_t* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (hs_gptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ u32x4 hs_ld16(hs_gptr base, uint32_t off) { return *(hs_gptr16)(base + off); }

constexpr int HS_C = 640, HS_H = 8, HS_D = 80, HS_TM = 64, HS_KS = HS_C / 16, HS_PW = 160;  // PW: features of a head pair / output columns of a quarter
constexpr int XROWB = HS_C * 2 + 16;   // X / O tile row stride: 81 sixteen-byte slots (odd: conflict-free ds_read_b128 over 32 rows)
constexpr int QROWB = HS_PW * 2 + 16;  // Q / K tile row stride: 21 slots
constexpr int VROWB = HS_TM * 2 + 8;   // V^T tile row stride: 34 dwords (conflict-free 8-byte reads over 32 rows)
constexpr int X_BYTES = HS_TM * XROWB, Q_BYTES = HS_TM * QROWB, V_BYTES = HS_PW * VROWB;

struct HsP {
    const uint8_t* x;
    const uint8_t* w;    // packed [4 pairs][NTILE][40 k-steps][64 lanes][8] (gamma and the softmax scale folded in)
    const float* wbias;  // [4 pairs][NTILE * 32] fp32: W . beta (+ the layer's bias), or nullptr
    const uint8_t* k1;
    const uint8_t* vt1;
    const float* bias1;
    const uint8_t* k2;
    const uint8_t* vt2;
    uint8_t* out;
    int32_t B, N, L1, Lpad1, L2, Lpad2, normalize;
    float eps, scale_log2, scale2;
};

// One sample's rows into the X tile: 8 lanes per row, 64 rows per pass of 512 threads.  NORM: (x - mean) * rstd, i.e. the LayerNorm
// WITHOUT its affine part -- gamma is folded into the packed weights and W . beta into their fp32 bias (ops.hs_pack_*; the same algebra as
// apad_gemm's folded LayerNorm, which this level's chain already uses): 20 parameter loads, 160 unpacks and 80 FMAs per lane less in
// the prologue every workgroup of a sample repeats.  The row stays packed in registers between the passes.
template <int DT, bool NORM>
__device__ __forceinline__ void hs_rows_to_lds(const uint8_t* xb, float eps, int nrows, uint8_t* X, int tid) {
    constexpr int CH = HS_C / 64;
    const int sub = tid & 7, row = tid >> 3;
    uint4 u[CH];
    const bool ok = row < nrows;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        u[i] = make_uint4(0u, 0u, 0u, 0u);
        if (ok) u[i] = *reinterpret_cast<const uint4*>(xb + ((int64_t)row * HS_C + (sub + 8 * i) * 8) * 2);
    }
    if (NORM) {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float v[8];
            unpack8<DT>(u[i], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s1 += v[e];
        }
        s1 += __shfl_xor(s1, 1);
        s1 += __shfl_xor(s1, 2);
        s1 += __shfl_xor(s1, 4);
        const float mean = s1 * (1.0f / HS_C);
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float v[8];
            unpack8<DT>(u[i], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dd = v[e] - mean;
                s2 = __builtin_fmaf(dd, dd, s2);
            }
        }
        s2 += __shfl_xor(s2, 1);
        s2 += __shfl_xor(s2, 2);
        s2 += __shfl_xor(s2, 4);
        const float rstd = ok ? rsqrtf(s2 * (1.0f / HS_C) + eps) : 0.f;
        const float nm = -mean * rstd;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            float v[8];
            unpack8<DT>(u[i], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaf(v[e], rstd, nm);
            *reinterpret_cast<uint4*>(X + row * XROWB + (sub + 8 * i) * 16) = pack8<DT>(v);
        }
    } else {
#pragma unroll
        for (int i = 0; i < CH; ++i) *reinterpret_cast<uint4*>(X + row * XROWB + (sub + 8 * i) * 16) = u[i];
    }
}

// acc[j][mt] += W_tile_j . X_panel_mt^T over the 40 k-steps; wf holds the first NSET k-steps' fragments on entry (requested by the caller
// before the LayerNorm); the fragment of k-step kk + NSET is requested right behind the MFMAs that consumed k-step kk, and the token
// fragments of k-step kk + 1 are read from LDS in front of the MFMAs of kk.  The order is PINNED with sched_group_barrier: left alone,
// hipcc sinks all 2 NSET loads of an unrolled body behind its last MFMA (prefetch distance 0: every body waits a full L2 round trip).
template <int DT, int NT, int NSET>
__device__ __forceinline__ void hs_project(const hs_gptr (&wb)[NT], uint32_t loff, const uint8_t* xs, typename ET<DT>::v8 (&wf)[NSET][NT], f32x16 (&acc)[NT][2]) {
    using E = ET<DT>;
    static_assert(HS_KS % NSET == 0, "the register sets of weight fragments rotate over the k-steps");
    typename E::v8 t[2][2];  // [k-step parity][panel]
#define HS_LDT(kk_, s_)                                                                             \
    t[s_][0] = as_v8<DT>(*reinterpret_cast<const uint4*>(xs + (kk_) * 32));                          \
    t[s_][1] = as_v8<DT>(*reinterpret_cast<const uint4*>(xs + 32 * XROWB + (kk_) * 32));
#define HS_MM(i_, s_)                                                                               \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                \
        if (!(HS_ABL & 4)) {                                                                        \
            acc[j][0] = E::mfma32(wf[i_][j], t[s_][0], acc[j][0]);                                  \
            acc[j][1] = E::mfma32(wf[i_][j], t[s_][1], acc[j][1]);                                  \
        }                                                                                           \
    }
    HS_LDT(0, 0);
#pragma unroll 1
    for (int kk = 0; kk < HS_KS - NSET; kk += NSET) {
#pragma unroll
        for (int i = 0; i < NSET; ++i) {
            HS_LDT(kk + i + 1, (i + 1) & 1);
            HS_MM(i, i & 1);
            if (!(HS_ABL & 1)) {
#pragma unroll
                for (int j = 0; j < NT; ++j) wf[i][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + (kk + i + NSET) * 1024, loff));
            }
        }
#pragma unroll
        for (int i = 0; i < NSET; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);           // 2 LDS reads (the NEXT k-step's token fragments)
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);      // the MFMAs of this k-step
            __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);          // the weight fragments NSET k-steps ahead
        }
    }
#pragma unroll
    for (int i = 0; i < NSET; ++i) {
        if (i + 1 < NSET) { HS_LDT(HS_KS - NSET + i + 1, (i + 1) & 1); }
        HS_MM(i, i & 1);
    }
#undef HS_LDT
#undef HS_MM
}

template <int DT, bool SELF, bool NORM, int NS1, int NS2, int NSET>
__global__ __launch_bounds__(512) void hs_attn_kernel(HsP p) {
    using E = ET<DT>;
    constexpr int D = HS_D, KC = D / 16, DTT = (D + 31) / 32;
    constexpr int NTILE = SELF ? 15 : 5, NT = SELF ? 2 : 1;
    constexpr bool DUAL = !SELF && NS2 > 0;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const X = smem;
    uint8_t* const Q = smem + X_BYTES;
    uint8_t* const K = Q + Q_BYTES;   // (self-attention only)
    uint8_t* const VT = K + Q_BYTES;  // (self-attention only)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x >> 2, pr = blockIdx.x & 3;
    const int N = p.N;

    HS_STAMP(0, 0);
    // ---- 0. the weight stream starts now: wave w owns row tiles w (and w + 8) of the pair's packed block ----
    const bool proj = SELF || wave < NTILE;
    hs_gptr wb[NT];
    typename E::v8 wf[NSET][NT];
    const uint32_t loff = (uint32_t)lane * 16u;
    int tile[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        tile[j] = wave + 8 * j;
        const int tl = tile[j] < NTILE ? tile[j] : NTILE - 1;  // (wave 7's second tile does not exist: it repeats the last one and drops the result)
        wb[j] = sgpr_ptr(p.w + ((int64_t)(pr * NTILE + tl) * HS_KS) * 1024);
    }
    if (proj) {
#pragma unroll
        for (int i = 0; i < NSET; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[i][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + i * 1024, loff));
    }

    // ---- 1. LayerNorm -> X ----
    HS_STAMP(0, 1);
    hs_rows_to_lds<DT, NORM>(p.x + (int64_t)b * N * HS_C * 2, p.eps, N, X, tid);
    HS_STAMP(0, 2);
    __syncthreads();
    HS_STAMP(0, 3);

    // ---- 2. projections -> Q (K, V^T) ----
    if (proj) {
        f32x16 acc[NT][2];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mt][r] = 0.f;
        hs_project<DT, NT, NSET>(wb, loff, X + l31 * XROWB + half * 16, wf, acc);
        HS_STAMP(0, 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = tile[j];
            if (p.wbias != nullptr && t < NTILE) {  // W . beta (+ bias): fp32, before the rounding
                const float* bp = p.wbias + (pr * NTILE + t) * 32 + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * g);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        acc[j][mt][4 * g + 0] += bv.x;
                        acc[j][mt][4 * g + 1] += bv.y;
                        acc[j][mt][4 * g + 2] += bv.z;
                        acc[j][mt][4 * g + 3] += bv.w;
                    }
                }
            }
            if (t < 10 && t < NTILE) {  // q / k: C layout = (lane: token, registers: 4 consecutive features) -> 8-byte stores into the row-major tile
                uint8_t* const dst = t < 5 ? Q : K;
                const int f0 = (t < 5 ? t : t - 5) * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        typename E::v4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = (typename E::elem)acc[j][mt][4 * g + e];
                        *reinterpret_cast<uint2*>(dst + (mt * 32 + l31) * QROWB + (f0 + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                    }
            } else if (SELF && t < NTILE) {  // v: the same C layout written transposed, V^T[feature][token] (2-byte stores, 64 contiguous bytes per half-wave)
                const int f0 = (t - 10) * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            *reinterpret_cast<typename E::elem*>(VT + (f0 + 8 * g + 4 * half + e) * VROWB + (mt * 32 + l31) * 2) = (typename E::elem)acc[j][mt][4 * g + e];
            }
        }
    }

    // ---- 3. attention: wave (h, panel), waves 0 .. 3.  Cross-attention: every K / V^T fragment of the head is requested before the barrier ----
    const int h = (wave >> 1) & 1, mt = wave & 1, hg = pr * 2 + h;
    constexpr bool BIG2 = NS2 > 2;  // the second segment's fragments are requested as they are used (short_segment_ns)
    constexpr bool SPLITF = DUAL && !BIG2 && NS1 + NS2 > 3;  // both resident sets would not fit: the second segment loads as it goes too
    constexpr int NSB = (DUAL && !BIG2 && !SPLITF) ? NS2 : 1;
    ShortFr<DT, D, NS1> f1;
    ShortFr<DT, D, NSB> f2;
    if (!SELF && wave < 4 && !(HS_ABL & 2)) {
        short_load<DT, D, NS1>(f1, p.k1 + ((int64_t)b * p.L1 * HS_C + hg * D) * 2, HS_C, p.vt1 + ((int64_t)(b * HS_H + hg) * D * p.Lpad1) * 2, p.L1, p.Lpad1, l31, half);
        if (DUAL && !BIG2 && !SPLITF)
            short_load<DT, D, NSB>(f2, p.k2 + ((int64_t)b * p.L2 * HS_C + hg * D) * 2, HS_C, p.vt2 + ((int64_t)(b * HS_H + hg) * D * p.Lpad2) * 2, p.L2, p.Lpad2, l31, half);
    }
    HS_STAMP(0, 5);
    __syncthreads();
    HS_STAMP(0, 6);
    if (wave >= 4 || (HS_ABL & 2)) {
        HS_STAMP(0, 9);
        return;
    }
    if (SELF) short_load<DT, D, NS1>(f1, K + h * D * 2, QROWB / 2, VT + h * D * VROWB, N, VROWB / 2, l31, half);
    typename E::v8 qf[KC];
    {
        const uint8_t* qp = Q + (mt * 32 + l31) * QROWB + (h * D + half * 8) * 2;
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) qf[cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(qp + cc * 32));
    }
    f32x16 o[DTT];
#pragma unroll
    for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float inv = 1.f;
    const float* const bias1 = (!SELF && p.bias1) ? p.bias1 + (int64_t)b * p.L1 : nullptr;
    short_compute<DT, D, NS1>(f1, SELF ? N : p.L1, bias1, p.scale_log2, qf, o, inv, half);
#pragma unroll
    for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= inv;
    if (DUAL) {
        f32x16 o2[DTT];
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o2[dt][r] = 0.f;
        float inv2 = 1.f;
        if constexpr (BIG2 || SPLITF)
            short_segment_ns<DT, D, (NS2 > 0 ? NS2 : 1)>(p.k2 + ((int64_t)b * p.L2 * HS_C + hg * D) * 2, HS_C, p.vt2 + ((int64_t)(b * HS_H + hg) * D * p.Lpad2) * 2, p.L2, p.Lpad2,
                                                          nullptr, p.scale_log2, qf, o2, inv2, l31, half);
        else
            short_compute<DT, D, NSB>(f2, p.L2, nullptr, p.scale_log2, qf, o2, inv2, half);
        // (as attn_short_kernel: each branch, and scale * audio, rounded to the storage type before the add)
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t = (float)(typename E::elem)o[dt][r];
                const float a = (float)(typename E::elem)(o2[dt][r] * inv2);
                o[dt][r] = t + (float)(typename E::elem)(p.scale2 * a);
            }
    }

    HS_STAMP(0, 7);
    // ---- 4. O(pair) -> HBM: lane = token, 8 bytes per (d-tile, group); the two halves of a wave write 16 contiguous bytes of a row ----
    const int row = mt * 32 + l31;
    if (row < N) {
        uint8_t* const ob = p.out + (((int64_t)b * N + row) * HS_C + hg * D) * 2;
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dcol = dt * 32 + 8 * g + 4 * half;
                if (dcol < D) {
                    typename E::v4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)o[dt][g * 4 + j];
                    *reinterpret_cast<uint2*>(ob + dcol * 2) = __builtin_bit_cast(uint2, pk);
                }
            }
    }
    HS_STAMP(0, 9);
}

struct HoP {
    const uint8_t* o;
    const uint8_t* w;  // packed [4 quarters][5 row tiles][40 k-steps][64 lanes][8]
    const uint8_t* bo;
    const uint8_t* res;
    uint8_t* out;
    float* rs_out;  // [B * N][20][2] (sum, sum of squares) of the stored row per 32-column tile, or nullptr
    int32_t B, N;
};

template <int DT, int NSET>
__global__ __launch_bounds__(512) void hs_out_kernel(HoP p) {
    using E = ET<DT>;
    constexpr int NTILE = 5;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const X = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x >> 2, cq = blockIdx.x & 3;
    const int N = p.N;
    HS_STAMP(1, 0);
    const bool proj = wave < NTILE;
    hs_gptr wb[1];
    typename E::v8 wf[NSET][1];
    const uint32_t loff = (uint32_t)lane * 16u;
    wb[0] = sgpr_ptr(p.w + ((int64_t)(cq * NTILE + (proj ? wave : 0)) * HS_KS) * 1024);
    const int c0 = cq * HS_PW + wave * 32;  // first output column of this wave's tile
    uint2 rx[2][4];
    if (proj) {
#pragma unroll
        for (int i = 0; i < NSET; ++i) wf[i][0] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[0] + i * 1024, loff));
        // the residual values of this wave's outputs, in the C layout (lane: token, 4 consecutive columns per group)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int row = mt * 32 + l31;
                rx[mt][g] = make_uint2(0u, 0u);
                if (row < N && p.res != nullptr) rx[mt][g] = *reinterpret_cast<const uint2*>(p.res + (((int64_t)b * N + row) * HS_C + c0 + 8 * g + 4 * half) * 2);
            }
    }
    HS_STAMP(1, 1);
    hs_rows_to_lds<DT, false>(p.o + (int64_t)b * N * HS_C * 2, 0.f, N, X, tid);
    HS_STAMP(1, 2);
    __syncthreads();
    HS_STAMP(1, 3);
    if (!proj) return;
    f32x16 acc[1][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][mt][r] = 0.f;
    hs_project<DT, 1, NSET>(wb, loff, X + l31 * XROWB + half * 16, wf, acc);
    HS_STAMP(1, 4);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = mt * 32 + l31;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = c0 + 8 * g + 4 * half;
            typename E::v4 rr = __builtin_bit_cast(typename E::v4, rx[mt][g]), y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float bv = p.bo != nullptr ? ld_elem<DT>(p.bo, col + e) : 0.f;
                const float lin = (float)(typename E::elem)(acc[0][mt][4 * g + e] + bv);  // to_out rounded, then the residual add rounded (the chain's two roundings)
                y[e] = (typename E::elem)(lin + (float)rr[e]);                              // (no residual: + 0 of an already rounded value is exact)
                const float yf = (float)y[e];
                s1 += yf;
                s2 = __builtin_fmaf(yf, yf, s2);
            }
            if (row < N) *reinterpret_cast<uint2*>(p.out + (((int64_t)b * N + row) * HS_C + col) * 2) = __builtin_bit_cast(uint2, y);
        }
        if (p.rs_out != nullptr) {
            s1 = half_sum(s1);
            s2 = half_sum(s2);
            if (half == 0 && row < N) *reinterpret_cast<float2*>(p.rs_out + (((int64_t)b * N + row) * 20 + cq * NTILE + wave) * 2) = make_float2(s1, s2);
        }
    }
    HS_STAMP(1, 9);
}

// dynamic LDS above 64 KB needs the attribute once per (kernel, device)
template <class K> int hs_ensure_lds(K kern, int bytes, bool (&done)[16], std::mutex& mu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    std::lock_guard<std::mutex> g(mu);
    if (!done[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
            apad_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed", bytes);
            return -1;
        }
        done[dev] = true;
    }
    return 0;
}

template <int DT, bool SELF, bool NORM, int NS1, int NS2> int hs_attn_go2(const HsP& p, hipStream_t s) {
    constexpr int NSET = HS_NSET;
    constexpr int LDS = SELF ? X_BYTES + 2 * Q_BYTES + V_BYTES : X_BYTES + Q_BYTES;
    static bool done[16] = {};
    static std::mutex mu;
    auto kern = hs_attn_kernel<DT, SELF, NORM, NS1, NS2, NSET>;
    if (hs_ensure_lds(kern, LDS, done, mu) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * 4)), dim3(512), LDS, s, p);
    return apad_check_launch("apad_hs_attention");
}
template <int DT, bool SELF, int NS1, int NS2> int hs_attn_go(const HsP& p, hipStream_t s) {
    return p.normalize ? hs_attn_go2<DT, SELF, true, NS1, NS2>(p, s) : hs_attn_go2<DT, SELF, false, NS1, NS2>(p, s);
}

template <int DT> int hs_attn_launch(const HsP& p, bool self, hipStream_t s) {
    if (self) return p.N > 32 ? hs_attn_go<DT, true, 2, 0>(p, s) : hs_attn_go<DT, true, 1, 0>(p, s);
    // sub-tile counts of the two segments are compile-time (the fragment registers of an unused sub-tile would not fit beside the rest)
    const int ns1 = p.L1 > 32 ? 2 : 1, ns2 = (p.L2 + 31) / 32;
    if (ns2 > 2) return hs_attn_go<DT, false, 1, 4>(p, s);  // (ns1 == 1: checked by the caller) 8 text + 65 .. 128 audio keys
    if (ns1 == 1 && ns2 == 0) return hs_attn_go<DT, false, 1, 0>(p, s);
    if (ns1 == 1 && ns2 == 1) return hs_attn_go<DT, false, 1, 1>(p, s);
    if (ns2 == 0) return hs_attn_go<DT, false, 2, 0>(p, s);
    return hs_attn_go<DT, false, 2, 2>(p, s);
}

template <int DT> int hs_out_launch(const HoP& p, hipStream_t s) {
    constexpr int NSET = HS_NSET;
    static bool done[16] = {};
    static std::mutex mu;
    auto kern = hs_out_kernel<DT, NSET>;
    if (hs_ensure_lds(kern, X_BYTES, done, mu) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * 4)), dim3(512), X_BYTES, s, p);
    return apad_check_launch("apad_hs_out");
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#ifdef HS_TRACE
extern "C" int apad_hs_trace_read(void* dst, int bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(hs_trace_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int apad_sizeof_hs_attn_desc(void) { return (int)sizeof(apad_hs_attn_desc); }
extern "C" int apad_sizeof_hs_out_desc(void) { return (int)sizeof(apad_hs_out_desc); }

extern "C" int apad_hs_attention(const apad_hs_attn_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_hs_attention: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_hs_attention: dtype %d not supported (16-bit only)", d->dtype);
    if (d->C != HS_C || d->heads != HS_H || d->N < 1 || d->N > HS_TM) {
        apad_set_error("apad_hs_attention: C=%d heads=%d N=%d outside the kernel envelope (640, 8, 1..64)", d->C, d->heads, d->N);
        return -3;
    }
    APAD_CHECK(d->x && d->w_packed && d->out, "apad_hs_attention: null operand");
    APAD_CHECK(d->B > 0, "apad_hs_attention: empty batch");
    const bool self = d->self_attention != 0;
    if (!self) {
        APAD_CHECK(d->k1 && d->vt1, "apad_hs_attention: cross-attention needs k1 / vt1");
        APAD_CHECK(d->L1 >= 1 && d->L1 <= 64 && d->L2 >= 0 && (d->L2 <= 64 || (d->L2 <= 128 && d->L1 <= 32)),
                   "apad_hs_attention: segment lengths %d / %d outside 1..64 / 0..64 (0..128 beside <= 32 keys in segment 1)", d->L1, d->L2);
        APAD_CHECK(d->Lpad1 >= d->L1 && d->Lpad1 % 32 == 0, "apad_hs_attention: Lpad1 must be >= L1 and a multiple of 32");
        if (d->L2 > 0) {
            APAD_CHECK(d->k2 && d->vt2, "apad_hs_attention: segment 2 needs k2 / vt2");
            APAD_CHECK(d->Lpad2 >= d->L2 && d->Lpad2 % 32 == 0, "apad_hs_attention: Lpad2 must be >= L2 and a multiple of 32");
        }
    } else {
        APAD_CHECK(d->key_bias == nullptr && d->L2 == 0, "apad_hs_attention: self-attention takes no key bias / second segment");
    }
    APAD_CHECK(al16(d->x) && al16(d->out) && al16(d->w_packed) && al16(d->k1) && al16(d->vt1) && al16(d->k2) && al16(d->vt2) && al16(d->w_bias),
               "apad_hs_attention: pointers must be 16-byte aligned");
    HsP p;
    p.x = (const uint8_t*)d->x; p.w = (const uint8_t*)d->w_packed; p.wbias = d->w_bias; p.normalize = d->normalize ? 1 : 0;
    p.k1 = (const uint8_t*)d->k1; p.vt1 = (const uint8_t*)d->vt1; p.bias1 = d->key_bias; p.k2 = (const uint8_t*)d->k2; p.vt2 = (const uint8_t*)d->vt2;
    p.out = (uint8_t*)d->out;
    p.B = d->B; p.N = d->N; p.L1 = d->L1; p.Lpad1 = d->Lpad1; p.L2 = d->L2; p.Lpad2 = d->Lpad2;
    p.eps = d->ln_eps; p.scale_log2 = d->q_prescaled ? 1.0f : d->softmax_scale * LOG2E; p.scale2 = d->scale2;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? hs_attn_launch<APAD_BF16>(p, self, s) : hs_attn_launch<APAD_F16>(p, self, s);
}

extern "C" int apad_hs_out(const apad_hs_out_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_hs_out: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_hs_out: dtype %d not supported (16-bit only)", d->dtype);
    if (d->C != HS_C || d->N < 1 || d->N > HS_TM) {
        apad_set_error("apad_hs_out: C=%d N=%d outside the kernel envelope (640, 1..64)", d->C, d->N);
        return -3;
    }
    APAD_CHECK(d->o && d->w_packed && d->out, "apad_hs_out: null operand");
    APAD_CHECK(d->B > 0, "apad_hs_out: empty batch");
    APAD_CHECK(al16(d->o) && al16(d->w_packed) && al16(d->residual) && al16(d->out) && (reinterpret_cast<uintptr_t>(d->rowstat_out) & 7) == 0,
               "apad_hs_out: pointers must be 16-byte aligned (rowstat_out: 8)");
    HoP p;
    p.o = (const uint8_t*)d->o; p.w = (const uint8_t*)d->w_packed; p.bo = (const uint8_t*)d->bias; p.res = (const uint8_t*)d->residual;
    p.out = (uint8_t*)d->out; p.rs_out = d->rowstat_out; p.B = d->B; p.N = d->N;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? hs_out_launch<APAD_BF16>(p, s) : hs_out_launch<APAD_F16>(p, s);
}
