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
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

#include "short_seg.h"

#ifndef HS_NSET
#define HS_NSET 8  // register sets of weight fragments = k-steps a fragment is requested ahead of its use (per wave: NSET x NT KB in flight)
#endif
#ifndef HS_PRE
#define HS_PRE 2  // of those, the sets requested BEFORE the sample's rows are normalised (the rest right after: a wave that is still pushing 16 KB
                  // of weight requests into the memory pipe cannot start on rows that have long arrived)
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
constexpr int X_BYTES = HS_TM * XROWB, Q_BYTES = HS_TM * QROWB, V_BYTES = HS_PW * VROWB, HS_BIAS_BYTES = 15 * 32 * 4;

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
    int32_t B, N, L1, Lpad1, L2, Lpad2, normalize, xm;
    float eps, scale_log2, scale2;
};

// One sample's rows into the X tile: 8 lanes per row, 64 rows per pass of 512 threads, in two steps so that the caller can put its weight
// requests BETWEEN them (vmcnt retires in order: rows requested behind 16 KB of weight fragments per wave would wait for those too).
// NORM: (x - mean) * rstd, i.e. the LayerNorm WITHOUT its affine part -- gamma is folded into the packed weights and W . beta into their
// fp32 bias (ops.hs_pack_*; the same algebra as apad_gemm's folded LayerNorm, which this level's chain already uses): 20 parameter loads,
// 160 unpacks and 80 FMAs per lane less in the prologue every workgroup of a sample repeats.  The row stays packed in registers.
constexpr int HS_CH = HS_C / 64;
__device__ __forceinline__ void hs_rows_load(uint4 (&u)[HS_CH], const uint8_t* xb, int nrows, int tid) {
    const int sub = tid & 7, row = tid >> 3;
    const int rr = row < nrows ? row : nrows - 1;  // (branch-free: a row past the sample re-reads its last row and is zeroed below)
#pragma unroll
    for (int i = 0; i < HS_CH; ++i) u[i] = *reinterpret_cast<const uint4*>(xb + ((int64_t)rr * HS_C + (sub + 8 * i) * 8) * 2);
}
template <int DT, bool NORM>
__device__ __forceinline__ void hs_rows_store(const uint4 (&u)[HS_CH], float eps, int nrows, uint8_t* X, int tid) {
    constexpr int CH = HS_CH;
    const int sub = tid & 7, row = tid >> 3;
    const bool ok = row < nrows;
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
        for (int i = 0; i < CH; ++i) *reinterpret_cast<uint4*>(X + row * XROWB + (sub + 8 * i) * 16) = ok ? u[i] : make_uint4(0u, 0u, 0u, 0u);
    }
}

// acc[j][mt] += W_tile_j . X_panel_mt^T over the 40 k-steps (SWAP1: slot 1 with the operands exchanged, acc[1][mt] = X_panel_mt . W_tile_1^T, so
// that its C layout is (lane: feature, registers: tokens) -- a V tile lands transposed without a transposing store).  wf holds the first NSET
// k-steps' fragments on entry; the fragment of k-step kk + NSET is requested right behind the MFMAs that consumed k-step kk, and the token
// fragments of k-step kk + 1 are read from LDS in front of the MFMAs of kk.  The order is PINNED with sched_group_barrier: left alone, hipcc
// sinks all 2 NSET loads of an unrolled body behind its last MFMA (prefetch distance 0: every body waits a full L2 round trip).
// CHAIN: a wave that runs several tile sets back to back (the GEGLU projection) requests the first NSET k-steps of the NEXT set (wbn) behind the
// last NSET k-steps of this one, so the stream does not drain under the epilogue in between.
template <int DT, int NT, int NSET, bool SWAP1, bool CHAIN = false>
__device__ __forceinline__ void hs_project(const hs_gptr (&wb)[NT], uint32_t loff, const uint8_t* xs, typename ET<DT>::v8 (&wf)[NSET][NT], f32x16 (&acc)[NT][2],
                                           const hs_gptr* wbn = nullptr) {
    using E = ET<DT>;
    static_assert(HS_KS % NSET == 0, "the register sets of weight fragments rotate over the k-steps");
    typename E::v8 t[2][2];  // [k-step parity][panel]
#define HS_LDT(kk_, s_)                                                                             \
    t[s_][0] = as_v8<DT>(*reinterpret_cast<const uint4*>(xs + (kk_) * 32));                          \
    t[s_][1] = as_v8<DT>(*reinterpret_cast<const uint4*>(xs + 32 * XROWB + (kk_) * 32));
#define HS_MM(i_, s_)                                                                               \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                \
        if (!(HS_ABL & 4)) {                                                                        \
            if (SWAP1 && j == 1) {                                                                  \
                acc[j][0] = E::mfma32(t[s_][0], wf[i_][j], acc[j][0]);                              \
                acc[j][1] = E::mfma32(t[s_][1], wf[i_][j], acc[j][1]);                              \
            } else {                                                                                \
                acc[j][0] = E::mfma32(wf[i_][j], t[s_][0], acc[j][0]);                              \
                acc[j][1] = E::mfma32(wf[i_][j], t[s_][1], acc[j][1]);                              \
            }                                                                                       \
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
        if (CHAIN) {
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[i][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wbn[j] + i * 1024, loff));
        }
    }
    if (CHAIN) {
#pragma unroll
        for (int i = 0; i < NSET; ++i) {
            if (i + 1 < NSET) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NT, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, NT, 0);
        }
    }
#undef HS_LDT
#undef HS_MM
}

// workgroup id -> (sample, quarter).  Workgroup i runs on XCD i % 8 (observed, speed only) and every launch has to pull its working set
// through its XCD's memory-side port (~1 TB/s per XCD: what the prologue of these kernels waits for), so the map decides how many bytes that
// is: with m sample classes on the XCD axis an XCD sees 1 / m of the samples' rows and m / 2 of the four weight quarters.  m = 2 (id = 4 b + q):
// one quarter, half of the rows; m = 8: all four quarters, an eighth of the rows.  Rows are 80 KB per sample, a quarter is 600 KB (q|k|v) or
// 200 KB (q / to_out): m = 4 for self-attention, 8 for the others.
__device__ __forceinline__ void hs_decode(int id, int m, int& b, int& q) {
    const int x = id & 7, r = id >> 3;
    const int pl = 8 / m, ph = 4 / pl;  // quarter classes on the XCD axis / on the remaining axis
    q = (r % ph) * pl + (x % pl);
    b = (r / ph) * m + (x / pl);
}
inline int hs_grid(int B, int m) { return 4 * m * ((B + m - 1) / m); }

// a [64][160] tile of 16-bit values in LDS (row stride QROWB) -> rows of a [.][640] tensor at column col0: 20 sixteen-byte chunks per row,
// consecutive lanes = consecutive chunks (whole 64-byte sectors per quad)
__device__ __forceinline__ void hs_tile_to_rows(const uint8_t* T, uint8_t* dst_rows, int col0, int nrows, int tid) {
    for (int idx = tid; idx < HS_TM * 20; idx += 512) {
        const int row = idx / 20, ch = idx - row * 20;
        if (row < nrows) *reinterpret_cast<uint4*>(dst_rows + ((int64_t)row * HS_C + col0 + ch * 8) * 2) = *reinterpret_cast<const uint4*>(T + row * QROWB + ch * 16);
    }
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
    uint8_t* const BIAS = Q + Q_BYTES;  // [NTILE * 32] fp32
    uint8_t* const K = BIAS + HS_BIAS_BYTES;  // (self-attention only)
    uint8_t* const VT = K + Q_BYTES;          // (self-attention only)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int b, pr;
    hs_decode(blockIdx.x, p.xm, b, pr);
    if (b >= p.B) return;  // (the grid is padded to whole XCD rounds)
    const int N = p.N;

    HS_STAMP(0, 0);
    // ---- 0. the sample's rows are requested first, the weight stream right behind them: wave w owns row tiles w (and w + 8) of the pair's block ----
    uint4 xr[HS_CH];
    hs_rows_load(xr, p.x + (int64_t)b * N * HS_C * 2, N, tid);
    const bool proj = SELF || wave < NTILE;
    const bool vslot = SELF && wave >= 2 && wave < 7;  // slot 1 of waves 2 .. 6 = tiles 10 .. 14 = the pair's V rows
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
    constexpr int PRE = HS_PRE < NSET ? HS_PRE : NSET;
    if (proj) {
#pragma unroll
        for (int i = 0; i < PRE; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[i][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + i * 1024, loff));
    }
    if (tid < NTILE * 8 && p.wbias != nullptr)  // the pair's fp32 bias -> LDS (read by the projection epilogue)
        *reinterpret_cast<float4*>(BIAS + tid * 16) = *reinterpret_cast<const float4*>(p.wbias + pr * NTILE * 32 + tid * 4);
    HS_STAMP(0, 1);

    // ---- 1. LayerNorm -> X ----
    hs_rows_store<DT, NORM>(xr, p.eps, N, X, tid);
    if (proj) {
#pragma unroll
        for (int i = PRE; i < NSET; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[i][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + i * 1024, loff));
    }
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
        if (vslot)
            hs_project<DT, NT, NSET, SELF>(wb, loff, X + l31 * XROWB + half * 16, wf, acc);
        else
            hs_project<DT, NT, NSET, false>(wb, loff, X + l31 * XROWB + half * 16, wf, acc);
        HS_STAMP(0, 4);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int t = tile[j];
            if (t >= NTILE) continue;
            const float* bp = p.wbias != nullptr ? reinterpret_cast<const float*>(BIAS) + t * 32 : nullptr;  // W . beta (+ bias): fp32 (staged in LDS), added before the rounding
            if (t < 10) {  // q / k: C layout = (lane: token, registers: 4 consecutive features) -> 8-byte stores into the row-major tile
                uint8_t* const dst = t < 5 ? Q : K;
                const int f0 = (t < 5 ? t : t - 5) * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bp != nullptr) bv = *reinterpret_cast<const float4*>(bp + 8 * g + 4 * half);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        typename E::v4 y;
                        y[0] = (typename E::elem)(acc[j][mt][4 * g + 0] + bv.x);
                        y[1] = (typename E::elem)(acc[j][mt][4 * g + 1] + bv.y);
                        y[2] = (typename E::elem)(acc[j][mt][4 * g + 2] + bv.z);
                        y[3] = (typename E::elem)(acc[j][mt][4 * g + 3] + bv.w);
                        *reinterpret_cast<uint2*>(dst + (mt * 32 + l31) * QROWB + (f0 + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                    }
                }
            } else if (SELF) {  // v (operands exchanged): C layout = (lane: feature, registers: 4 consecutive tokens) -> 8-byte stores into V^T[feature][token]
                const int f = (t - 10) * 32 + l31;
                const float bv = bp != nullptr ? bp[l31] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        typename E::v4 y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) y[e] = (typename E::elem)(acc[j][mt][4 * g + e] + bv);
                        *reinterpret_cast<uint2*>(VT + f * VROWB + (mt * 32 + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                    }
            }
        }
    }

    // ---- 3. attention: wave (h, panel), waves 0 .. 3.  Cross-attention: every K / V^T fragment of the head is requested before the barrier ----
    const int h = (wave >> 1) & 1, mt = wave & 1, hg = pr * 2 + h;
    constexpr bool BIG2 = NS2 > 2;  // the second segment's fragments are requested as they are used (short_segment_ns)
    constexpr bool LONG2 = NS2 > 4;  // ... in 64-key chunks with a running maximum / sum (long_segment: 129 .. 512 audio keys)
    constexpr bool SPLITF = DUAL && !BIG2 && NS1 + NS2 > 3;  // both resident sets would not fit: the second segment loads as it goes too
    constexpr int NSB = (DUAL && !BIG2 && !SPLITF) ? NS2 : 1;
    const bool att = wave < 4 && !(HS_ABL & 2);
    ShortFr<DT, D, NS1> f1;
    ShortFr<DT, D, NSB> f2;
    // the fragment sources of the head's two segments: fragment-packed sets (vtN == nullptr: apad_rows_pack_kv, round 6) or apad_attention's tensors
    const bool pk1 = !SELF && p.vt1 == nullptr, pk2 = DUAL && p.vt2 == nullptr;
    const int64_t hb1 = kv_packed_head_bytes(D, SELF ? 32 : p.L1), hb2 = kv_packed_head_bytes(D, DUAL ? p.L2 : 32);
#define HS_RAW1 KvRaw<DT, D>{p.k1 + ((int64_t)b * p.L1 * HS_C + hg * D) * 2, HS_C, p.vt1 + ((int64_t)(b * HS_H + hg) * D * p.Lpad1) * 2, p.L1, p.Lpad1}
#define HS_RAW2 KvRaw<DT, D>{p.k2 + ((int64_t)b * p.L2 * HS_C + hg * D) * 2, HS_C, p.vt2 + ((int64_t)(b * HS_H + hg) * D * p.Lpad2) * 2, p.L2, p.Lpad2}
#define HS_PK1 KvPacked<DT, D>{p.k1 + (int64_t)(b * HS_H + hg) * hb1, p.L1, (p.L1 + 31) & ~31}
#define HS_PK2 KvPacked<DT, D>{p.k2 + (int64_t)(b * HS_H + hg) * hb2, p.L2, (p.L2 + 31) & ~31}
    if (!SELF && att) {
        if (pk1) short_load<DT, D, NS1>(f1, HS_PK1, l31, half);
        else short_load<DT, D, NS1>(f1, HS_RAW1, l31, half);
        if (DUAL && !BIG2 && !SPLITF) {
            if (pk2) short_load<DT, D, NSB>(f2, HS_PK2, l31, half);
            else short_load<DT, D, NSB>(f2, HS_RAW2, l31, half);
        }
    }
    HS_STAMP(0, 5);
    __syncthreads();
    HS_STAMP(0, 6);
    if (att) {
        if (SELF) short_load<DT, D, NS1>(f1, KvRaw<DT, D>{K + h * D * 2, QROWB / 2, VT + h * D * VROWB, N, VROWB / 2}, l31, half);
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
            if constexpr (LONG2) {
                // (splitting the chunks over the four waves that idle in this phase was measured: no change -- at 64 samples x 8 heads x 520 keys the launch moves
                //  its 85 MB of key / value sets at 5.8 TB/s)
                if (pk2) long_segment<DT, D>(HS_PK2, p.scale_log2, qf, o2, inv2, l31, half);
                else long_segment<DT, D>(HS_RAW2, p.scale_log2, qf, o2, inv2, l31, half);
            } else if constexpr (BIG2 || SPLITF) {
                constexpr int NSX = NS2 > 0 ? NS2 : 1;
                if (pk2) short_segment_ns<DT, D, NSX>(HS_PK2, nullptr, p.scale_log2, qf, o2, inv2, l31, half);
                else short_segment_ns<DT, D, NSX>(HS_RAW2, nullptr, p.scale_log2, qf, o2, inv2, l31, half);
            }
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
        // O(head, panel) over this wave's own q values in the Q tile (nobody else reads them)
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dcol = dt * 32 + 8 * g + 4 * half;
                if (dcol < D) {
                    typename E::v4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)o[dt][g * 4 + j];
                    *reinterpret_cast<uint2*>(Q + (mt * 32 + l31) * QROWB + (h * D + dcol) * 2) = __builtin_bit_cast(uint2, pk);
                }
            }
    }
#undef HS_RAW1
#undef HS_RAW2
#undef HS_PK1
#undef HS_PK2
    __syncthreads();
    // ---- 4. O(pair) [64][160] -> HBM, whole 16-byte chunks ----
    hs_tile_to_rows(Q, p.out + (int64_t)b * N * HS_C * 2, pr * HS_PW, N, tid);
    HS_STAMP(0, 9);
}

// ---- the feed-forward's first half at the same level: H = value * gelu(gate), [value | gate] = Linear(LayerNorm(x)) (diffusers GEGLU: proj 640 -> 5120) ----
// workgroup = (sample, hidden quarter): the 640 hidden units of the quarter are 20 tiles of 32, each a (value, gate) pair of weight row tiles;
// wave w runs hidden tiles w, w + 8 (, w + 16 for w < 4: the two waves of a SIMD together always 5) one after the other against the X tile, the weight
// stream of the next tile requested under the tail of the current one; value and gate rounded like the chain's Linear output, the product rounded once;
// H rows leave through a per-wave [64][32] LDS scratch as 64-byte row segments.
struct HgP {
    const uint8_t* x;
    const uint8_t* w;    // packed [4 quarters][20 hidden tiles][2: value, gate][40 k-steps][64 lanes][8] (gamma folded in)
    const float* wbias;  // [4][20][2][32] fp32: W . beta + bias
    uint8_t* out;        // H [B * N][2560]
    int32_t B, N, normalize, xm;
    float eps;
};
constexpr int HG_TILES = 20, HG_SROWB = 32 * 2 + 16, HG_STG = HS_TM * HG_SROWB, HG_BIAS_BYTES = HG_TILES * 2 * 32 * 4;

template <int DT, bool NORM, int NSET>
__global__ __launch_bounds__(512) void hs_geglu_kernel(HgP p) {
    using E = ET<DT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const X = smem;
    uint8_t* const BIAS = smem + X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* const STG = BIAS + HG_BIAS_BYTES + wave * HG_STG;
    const int half = lane >> 5, l31 = lane & 31;
    int b, hq;
    hs_decode(blockIdx.x, p.xm, b, hq);
    if (b >= p.B) return;
    const int N = p.N;
    uint4 xr[HS_CH];
    hs_rows_load(xr, p.x + (int64_t)b * N * HS_C * 2, N, tid);
    const int ntl = wave < 4 ? 3 : 2;
    const uint32_t loff = (uint32_t)lane * 16u;
    auto tile_base = [&](int t, int j) { return sgpr_ptr(p.w + ((int64_t)((hq * HG_TILES + t) * 2 + j) * HS_KS) * 1024); };
    hs_gptr wb[2] = {tile_base(wave, 0), tile_base(wave, 1)};
    typename E::v8 wf[NSET][2];
    constexpr int PRE = HS_PRE < NSET ? HS_PRE : NSET;
#pragma unroll
    for (int i = 0; i < PRE; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[i][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + i * 1024, loff));
    if (tid < HG_TILES * 2 * 8 && p.wbias != nullptr)
        *reinterpret_cast<float4*>(BIAS + tid * 16) = *reinterpret_cast<const float4*>(p.wbias + hq * HG_TILES * 2 * 32 + tid * 4);
    hs_rows_store<DT, NORM>(xr, p.eps, N, X, tid);
#pragma unroll
    for (int i = PRE; i < NSET; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) wf[i][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + i * 1024, loff));
    __syncthreads();
    uint8_t* const hb = p.out + ((int64_t)b * N * 4 * HS_C + hq * HS_C) * 2;
#pragma unroll 1
    for (int it = 0; it < ntl; ++it) {
        const int t = wave + 8 * it;
        const int tn = it + 1 < ntl ? t + 8 : t;  // (after the last tile: its own first fragments once more, unused)
        const hs_gptr wbn[2] = {tile_base(tn, 0), tile_base(tn, 1)};
        f32x16 acc[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][mt][r] = 0.f;
        hs_project<DT, 2, NSET, false, true>(wb, loff, X + l31 * XROWB + half * 16, wf, acc, wbn);
        wb[0] = wbn[0];
        wb[1] = wbn[1];
        const float* bp = reinterpret_cast<const float*>(BIAS) + t * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
            if (p.wbias != nullptr) {
                bv = *reinterpret_cast<const float4*>(bp + 8 * g + 4 * half);
                bg = *reinterpret_cast<const float4*>(bp + 32 + 8 * g + 4 * half);
            }
            const float bva[4] = {bv.x, bv.y, bv.z, bv.w}, bga[4] = {bg.x, bg.y, bg.z, bg.w};
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                typename E::v4 y;
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // (value and gate rounded to the storage type like the chain's Linear output; the product rounded once)
                    const apad_f32x2 gg = {(float)(typename E::elem)(acc[1][mt][4 * g + e] + bga[e]), (float)(typename E::elem)(acc[1][mt][4 * g + e + 1] + bga[e + 1])};
                    const apad_f32x2 ge = gelu_erf_2(gg);
                    y[e] = (typename E::elem)((float)(typename E::elem)(acc[0][mt][4 * g + e] + bva[e]) * ge[0]);
                    y[e + 1] = (typename E::elem)((float)(typename E::elem)(acc[0][mt][4 * g + e + 1] + bva[e + 1]) * ge[1]);
                }
                *reinterpret_cast<uint2*>(STG + (mt * 32 + l31) * HG_SROWB + (8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
            }
        }
        // this wave's [64][32] tile -> H: 4 lanes per row, 64 contiguous bytes per row and instruction
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int row = q4 * 16 + (lane >> 2), ch = lane & 3;
            const uint4 v = *reinterpret_cast<const uint4*>(STG + row * HG_SROWB + ch * 16);
            if (row < N) *reinterpret_cast<uint4*>(hb + ((int64_t)row * 4 * HS_C + t * 32 + ch * 8) * 2) = v;
        }
    }
}

struct HoP {
    const uint8_t* o;
    const uint8_t* w;  // packed [4 quarters][5 row tiles][40 k-steps][64 lanes][8]
    const uint8_t* bo;
    const uint8_t* res;
    uint8_t* out;
    float* rs_out;  // [B * N][20][2] (sum, sum of squares) of the stored row per 32-column tile, or nullptr
    int32_t B, N, xm;
};

__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
    return v;
}

template <int DT, int NSET>
__global__ __launch_bounds__(512) void hs_out_kernel(HoP p) {
    using E = ET<DT>;
    constexpr int NTILE = 5;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const X = smem;
    uint8_t* const T = smem + X_BYTES;  // [64][160] to_out (+ bias), rounded
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    int b, cq;
    hs_decode(blockIdx.x, p.xm, b, cq);
    if (b >= p.B) return;
    const int N = p.N;
    HS_STAMP(1, 0);
    // the O rows first, then this wave's weight stream, then what the epilogue needs (bias in the C layout, the residual chunks of the
    // final coalesced pass): everything is in flight before the first wait
    uint4 xr[HS_CH];
    hs_rows_load(xr, p.o + (int64_t)b * N * HS_C * 2, N, tid);
    const bool proj = wave < NTILE;
    hs_gptr wb[1];
    typename E::v8 wf[NSET][1];
    const uint32_t loff = (uint32_t)lane * 16u;
    wb[0] = sgpr_ptr(p.w + ((int64_t)(cq * NTILE + (proj ? wave : 0)) * HS_KS) * 1024);
    uint2 bq[4];
    constexpr int PRE = HS_PRE < NSET ? HS_PRE : NSET;
    if (proj) {
#pragma unroll
        for (int i = 0; i < PRE; ++i) wf[i][0] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[0] + i * 1024, loff));
    }
    HS_STAMP(1, 1);
    hs_rows_store<DT, false>(xr, 0.f, N, X, tid);
    if (proj) {
#pragma unroll
        for (int i = PRE; i < NSET; ++i) wf[i][0] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[0] + i * 1024, loff));
#pragma unroll
        for (int g = 0; g < 4; ++g) bq[g] = p.bo != nullptr ? *reinterpret_cast<const uint2*>(p.bo + (cq * HS_PW + wave * 32 + 8 * g + 4 * half) * 2) : make_uint2(0u, 0u);
    }
    constexpr int NPASS = (HS_TM * 20 + 511) / 512;
    uint4 rres[NPASS];
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
        const int idx = tid + it * 512, row = idx / 20, ch = idx - row * 20;
        rres[it] = make_uint4(0u, 0u, 0u, 0u);
        if (p.res != nullptr && row < N) rres[it] = *reinterpret_cast<const uint4*>(p.res + (((int64_t)b * N + row) * HS_C + cq * HS_PW + ch * 8) * 2);
    }
    HS_STAMP(1, 2);
    __syncthreads();
    HS_STAMP(1, 3);
    if (proj) {
        f32x16 acc[1][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][mt][r] = 0.f;
        hs_project<DT, 1, NSET, false>(wb, loff, X + l31 * XROWB + half * 16, wf, acc);
        HS_STAMP(1, 4);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const typename E::v4 bv = __builtin_bit_cast(typename E::v4, bq[g]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                typename E::v4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (typename E::elem)(acc[0][mt][4 * g + e] + (float)bv[e]);  // to_out + bias, rounded (the chain's first rounding)
                *reinterpret_cast<uint2*>(T + (mt * 32 + l31) * QROWB + (wave * 32 + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
            }
        }
    }
    __syncthreads();
    // + residual (the second rounding), whole 16-byte chunks: lanes 4 k .. 4 k + 3 hold one 32-column tile of one row -> its statistics by two DPP steps
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
        const int idx = tid + it * 512, row = idx / 20, ch = idx - row * 20;
        float y[8], r[8];
        float s1 = 0.f, s2 = 0.f;
        if (idx < HS_TM * 20) {
            unpack8<DT>(*reinterpret_cast<const uint4*>(T + row * QROWB + ch * 16), y);
            unpack8<DT>(rres[it], r);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = r[e] = 0.f;
        }
        const uint4 pk = [&] {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] += r[e];
            return pack8<DT>(y);
        }();
        if (p.rs_out != nullptr) {
            float z[8];
            unpack8<DT>(pk, z);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += z[e];
                s2 = __builtin_fmaf(z[e], z[e], s2);
            }
            s1 = quad_sum(s1);
            s2 = quad_sum(s2);
        }
        if (idx < HS_TM * 20 && row < N) {
            *reinterpret_cast<uint4*>(p.out + (((int64_t)b * N + row) * HS_C + cq * HS_PW + ch * 8) * 2) = pk;
            if (p.rs_out != nullptr && (lane & 3) == 0) *reinterpret_cast<float2*>(p.rs_out + (((int64_t)b * N + row) * 20 + cq * NTILE + (ch >> 2)) * 2) = make_float2(s1, s2);
        }
    }
    HS_STAMP(1, 9);
}

// ---- the feed-forward's second half: out = x + (H . W2^T + b2), H [64][2560] per sample, workgroup = (sample, output-column quarter) ----
// K = 2560 does not fit LDS: H streams through two [64][320] LDS buffers (8 chunks of 20 k-steps; chunk c + 1 is fetched into registers while chunk c
// is multiplied, stored behind it, one barrier per chunk).  The 8 waves split every chunk by K-QUARTER x TILE GROUP -- wave (kq, tg) runs k-steps
// 5 kq .. 5 kq + 4 of the chunk for row tiles {0, 1, 2} (tg 0) or {3, 4} (tg 1) on both token panels -- so every packed weight fragment is loaded exactly
// once per workgroup (5 k-steps = one chunk ahead) and the two waves of a SIMD (w, w + 4) together always issue 50 MFMAs per chunk.  The four K-quarter
// partials of an output tile are summed in a FIXED order (kq 0 + 1 + 2 + 3, through LDS), then bias, rounding, + residual and the row statistics exactly
// as in hs_out_kernel.
struct HfP {
    const uint8_t* h;    // [B * N][2560]
    const uint8_t* w;    // packed [4 quarters][5 row tiles][160 k-steps][64 lanes][8]
    const uint8_t* bo;
    const uint8_t* res;
    uint8_t* out;
    float* rs_out;
    int32_t B, N, xm;
};
constexpr int HF_K = 4 * HS_C, HF_KS = HF_K / 16, HF_CK = 20, HF_NCH = HF_KS / HF_CK, HF_ROWB = HF_CK * 32 + 16, HF_BUF = HS_TM * HF_ROWB;
constexpr int HF_RED = 3 * 10 * 4096;  // fp32 partials of the K-quarters 1 .. 3: [kq - 1][tile * 2 + panel][4 register groups][64 lanes][4]
constexpr int HF_LDS = (2 * HF_BUF > HF_RED ? 2 * HF_BUF : HF_RED) + Q_BYTES;

template <int DT, int NTW>
__device__ __forceinline__ void hf_body(const HfP& p, uint8_t* smem, int b, int cq, int tid, int lane, int wave) {
    using E = ET<DT>;
    const int half = lane >> 5, l31 = lane & 31, kq = wave & 3, t0 = (wave >> 2) * 3;
    const int N = p.N;
    uint8_t* const T = smem + (HF_LDS - Q_BYTES);
    const uint32_t loff = (uint32_t)lane * 16u;
    hs_gptr wb[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) wb[j] = sgpr_ptr(p.w + ((int64_t)((cq * 5 + t0 + j) * HF_KS + kq * 5)) * 1024);
    // H chunk staging: 64 rows x 40 sixteen-byte pieces = 5 per thread
    const uint8_t* const hb = p.h + (int64_t)b * N * HF_K * 2;
    int srow[5], sch[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int idx = tid + i * 512;
        srow[i] = idx / 40;
        sch[i] = idx - srow[i] * 40;
    }
    uint4 hr[5];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int rr = srow[i] < N ? srow[i] : N - 1;
            hr[i] = *reinterpret_cast<const uint4*>(hb + ((int64_t)rr * HF_K + c * (HF_CK * 16) + sch[i] * 8) * 2);
        }
    };
    auto store_chunk = [&](int c) {
        uint8_t* const dst = smem + (c & 1) * HF_BUF;
#pragma unroll
        for (int i = 0; i < 5; ++i) *reinterpret_cast<uint4*>(dst + srow[i] * HF_ROWB + sch[i] * 16) = srow[i] < N ? hr[i] : make_uint4(0u, 0u, 0u, 0u);
    };
    load_chunk(0);
    typename E::v8 wf[5][NTW];
#pragma unroll
    for (int s_ = 0; s_ < 5; ++s_)
#pragma unroll
        for (int j = 0; j < NTW; ++j) wf[s_][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + s_ * 1024, loff));
    uint2 bq[NTW][4];
    if (kq == 0) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bq[j][g] = p.bo != nullptr ? *reinterpret_cast<const uint2*>(p.bo + (cq * HS_PW + (t0 + j) * 32 + 8 * g + 4 * half) * 2) : make_uint2(0u, 0u);
    }
    constexpr int NPASS = (HS_TM * 20 + 511) / 512;
    uint4 rres[NPASS];
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
        const int idx = tid + it * 512, row = idx / 20, ch = idx - row * 20;
        rres[it] = make_uint4(0u, 0u, 0u, 0u);
        if (p.res != nullptr && row < N) rres[it] = *reinterpret_cast<const uint4*>(p.res + (((int64_t)b * N + row) * HS_C + cq * HS_PW + ch * 8) * 2);
    }
    store_chunk(0);
    __syncthreads();
    f32x16 acc[NTW][2];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][mt][r] = 0.f;
#pragma unroll 1
    for (int c = 0; c < HF_NCH; ++c) {
        const bool more = c + 1 < HF_NCH;
        const int cn = more ? c + 1 : c;  // (last chunk: re-reads itself, unused -- keeps the loop branch-free)
        load_chunk(cn);
        const uint8_t* xs = smem + (c & 1) * HF_BUF + l31 * HF_ROWB + half * 16 + kq * 5 * 32;
#pragma unroll
        for (int s_ = 0; s_ < 5; ++s_) {
            const typename E::v8 ta = as_v8<DT>(*reinterpret_cast<const uint4*>(xs + s_ * 32));
            const typename E::v8 tb = as_v8<DT>(*reinterpret_cast<const uint4*>(xs + 32 * HF_ROWB + s_ * 32));
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                acc[j][0] = E::mfma32(wf[s_][j], ta, acc[j][0]);
                acc[j][1] = E::mfma32(wf[s_][j], tb, acc[j][1]);
            }
#pragma unroll
            for (int j = 0; j < NTW; ++j) wf[s_][j] = __builtin_bit_cast(typename E::v8, hs_ld16(wb[j] + (cn * HF_CK + s_) * 1024, loff));
        }
#pragma unroll
        for (int s_ = 0; s_ < 5; ++s_) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTW, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, NTW, 0);
        }
        store_chunk(c + 1);  // (c + 1 == HF_NCH lands in the buffer nobody reads any more)
        __syncthreads();
    }
    // ---- K-quarter partials -> one sum per tile, fixed order ----
    if (kq != 0) {
        float* const R = reinterpret_cast<float*>(smem) + (kq - 1) * 10 * 1024;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(R + (((t0 + j) * 2 + mt) * 4 + g) * 256 + lane * 4) =
                        make_float4(acc[j][mt][4 * g], acc[j][mt][4 * g + 1], acc[j][mt][4 * g + 2], acc[j][mt][4 * g + 3]);
    }
    __syncthreads();
    if (kq == 0) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const typename E::v4 bv = __builtin_bit_cast(typename E::v4, bq[j][g]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float v[4] = {acc[j][mt][4 * g], acc[j][mt][4 * g + 1], acc[j][mt][4 * g + 2], acc[j][mt][4 * g + 3]};
#pragma unroll
                    for (int k_ = 0; k_ < 3; ++k_) {
                        const float4 o = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem) + k_ * 10 * 1024 + (((t0 + j) * 2 + mt) * 4 + g) * 256 + lane * 4);
                        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                    }
                    typename E::v4 y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = (typename E::elem)(v[e] + (float)bv[e]);
                    *reinterpret_cast<uint2*>(T + (mt * 32 + l31) * QROWB + ((t0 + j) * 32 + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                }
            }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NPASS; ++it) {
        const int idx = tid + it * 512, row = idx / 20, ch = idx - row * 20;
        float y[8], r[8];
        float s1 = 0.f, s2 = 0.f;
        if (idx < HS_TM * 20) {
            unpack8<DT>(*reinterpret_cast<const uint4*>(T + row * QROWB + ch * 16), y);
            unpack8<DT>(rres[it], r);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = r[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] += r[e];
        const uint4 pk = pack8<DT>(y);
        if (p.rs_out != nullptr) {
            float z[8];
            unpack8<DT>(pk, z);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1 += z[e];
                s2 = __builtin_fmaf(z[e], z[e], s2);
            }
            s1 = quad_sum(s1);
            s2 = quad_sum(s2);
        }
        if (idx < HS_TM * 20 && row < N) {
            *reinterpret_cast<uint4*>(p.out + (((int64_t)b * N + row) * HS_C + cq * HS_PW + ch * 8) * 2) = pk;
            if (p.rs_out != nullptr && (lane & 3) == 0) *reinterpret_cast<float2*>(p.rs_out + (((int64_t)b * N + row) * 20 + cq * 5 + (ch >> 2)) * 2) = make_float2(s1, s2);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(512) void hs_ff2_kernel(HfP p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, cq;
    hs_decode(blockIdx.x, p.xm, b, cq);
    if (b >= p.B) return;
    if (wave < 4)
        hf_body<DT, 3>(p, smem, b, cq, tid, lane, wave);
    else
        hf_body<DT, 2>(p, smem, b, cq, tid, lane, wave);
}

template <int DT, bool SELF, bool NORM, int NS1, int NS2> int hs_attn_go2(const HsP& p, hipStream_t s) {
    constexpr int NSET = HS_NSET;
    constexpr int LDS = SELF ? X_BYTES + 2 * Q_BYTES + V_BYTES + HS_BIAS_BYTES : X_BYTES + Q_BYTES + HS_BIAS_BYTES;
    static unsigned devs = 0;
    auto kern = hs_attn_kernel<DT, SELF, NORM, NS1, NS2, NSET>;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)hs_grid(p.B, p.xm)), dim3(512), LDS, s, p);
    return apad_check_launch("apad_hs_attention");
}
template <int DT, bool SELF, int NS1, int NS2> int hs_attn_go(const HsP& p, hipStream_t s) {
    return p.normalize ? hs_attn_go2<DT, SELF, true, NS1, NS2>(p, s) : hs_attn_go2<DT, SELF, false, NS1, NS2>(p, s);
}

template <int DT> int hs_attn_launch(const HsP& p, bool self, hipStream_t s) {
    if (self) return p.N > 32 ? hs_attn_go<DT, true, 2, 0>(p, s) : hs_attn_go<DT, true, 1, 0>(p, s);
    // sub-tile counts of the two segments are compile-time (the fragment registers of an unused sub-tile would not fit beside the rest)
    const int ns1 = p.L1 > 32 ? 2 : 1, ns2 = (p.L2 + 31) / 32;
    if (ns2 > 2) return hs_attn_go<DT, false, 1, 16>(p, s);  // (ns1 == 1: checked by the caller) 8 text + 65 .. 512 audio keys, in 64-key chunks (long_segment)
    if (ns1 == 1 && ns2 == 0) return hs_attn_go<DT, false, 1, 0>(p, s);
    if (ns1 == 1 && ns2 == 1) return hs_attn_go<DT, false, 1, 1>(p, s);
    if (ns2 == 0) return hs_attn_go<DT, false, 2, 0>(p, s);
    return hs_attn_go<DT, false, 2, 2>(p, s);
}

template <int DT> int hs_out_launch(const HoP& p, hipStream_t s) {
    constexpr int NSET = HS_NSET;
    static unsigned devs = 0;
    auto kern = hs_out_kernel<DT, NSET>;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), X_BYTES + Q_BYTES, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)hs_grid(p.B, p.xm)), dim3(512), X_BYTES + Q_BYTES, s, p);
    return apad_check_launch("apad_hs_out");
}

template <int DT, bool NORM> int hs_geglu_go(const HgP& p, hipStream_t s) {
    constexpr int LDS = X_BYTES + HG_BIAS_BYTES + 8 * HG_STG;
    static unsigned devs = 0;
    auto kern = hs_geglu_kernel<DT, NORM, HS_NSET>;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)hs_grid(p.B, p.xm)), dim3(512), LDS, s, p);
    return apad_check_launch("apad_hs_geglu");
}

template <int DT> int hs_ff2_launch(const HfP& p, hipStream_t s) {
    static unsigned devs = 0;
    auto kern = hs_ff2_kernel<DT>;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), HF_LDS, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)hs_grid(p.B, p.xm)), dim3(512), HF_LDS, s, p);
    return apad_check_launch("apad_hs_ff2");
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
        APAD_CHECK(d->k1, "apad_hs_attention: cross-attention needs k1 / vt1 (or the segment's apad_rows_pack_kv set in k1, vt1 = NULL)");
        APAD_CHECK(d->L1 >= 1 && d->L1 <= 64 && d->L2 >= 0 && (d->L2 <= 64 || (d->L2 <= 512 && d->L1 <= 32)),
                   "apad_hs_attention: segment lengths %d / %d outside 1..64 / 0..64 (0..512 beside <= 32 keys in segment 1)", d->L1, d->L2);
        APAD_CHECK(d->vt1 == nullptr || (d->Lpad1 >= d->L1 && d->Lpad1 % 32 == 0), "apad_hs_attention: Lpad1 must be >= L1 and a multiple of 32");
        if (d->L2 > 0) {
            APAD_CHECK(d->k2, "apad_hs_attention: segment 2 needs k2 / vt2 (or its packed set in k2)");
            APAD_CHECK(d->vt2 == nullptr || (d->Lpad2 >= d->L2 && d->Lpad2 % 32 == 0), "apad_hs_attention: Lpad2 must be >= L2 and a multiple of 32");
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
    static const int xm_self = 8, xm_cross = 8;  // (A/B knobs, read once)
    p.xm = self ? xm_self : xm_cross;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? hs_attn_launch<APAD_BF16>(p, self, s) : hs_attn_launch<APAD_F16>(p, self, s);
}

extern "C" int apad_hs_geglu(const void* x, const void* w_packed, const float* w_bias, void* out, int32_t B, int32_t N, int32_t C, int32_t normalize, float ln_eps,
                             int32_t dtype, void* stream) {
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_hs_geglu: dtype %d not supported (16-bit only)", dtype);
    if (C != HS_C || N < 1 || N > HS_TM) {
        apad_set_error("apad_hs_geglu: C=%d N=%d outside the kernel envelope (640, 1..64)", C, N);
        return -3;
    }
    APAD_CHECK(x && w_packed && out && B > 0, "apad_hs_geglu: null operand / empty batch");
    APAD_CHECK(al16(x) && al16(w_packed) && al16(out) && al16(w_bias), "apad_hs_geglu: pointers must be 16-byte aligned");
    HgP p;
    p.x = (const uint8_t*)x; p.w = (const uint8_t*)w_packed; p.wbias = w_bias; p.out = (uint8_t*)out;
    p.B = B; p.N = N; p.normalize = normalize ? 1 : 0; p.eps = ln_eps;
    static const int xm = 2;
    p.xm = xm;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_BF16) return p.normalize ? hs_geglu_go<APAD_BF16, true>(p, s) : hs_geglu_go<APAD_BF16, false>(p, s);
    return p.normalize ? hs_geglu_go<APAD_F16, true>(p, s) : hs_geglu_go<APAD_F16, false>(p, s);
}

extern "C" int apad_hs_ff2(const apad_hs_out_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_hs_ff2: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_hs_ff2: dtype %d not supported (16-bit only)", d->dtype);
    if (d->C != HS_C || d->N < 1 || d->N > HS_TM) {
        apad_set_error("apad_hs_ff2: C=%d N=%d outside the kernel envelope (640, 1..64)", d->C, d->N);
        return -3;
    }
    APAD_CHECK(d->o && d->w_packed && d->out && d->B > 0, "apad_hs_ff2: null operand / empty batch");
    APAD_CHECK(al16(d->o) && al16(d->w_packed) && al16(d->residual) && al16(d->out) && (reinterpret_cast<uintptr_t>(d->rowstat_out) & 7) == 0,
               "apad_hs_ff2: pointers must be 16-byte aligned (rowstat_out: 8)");
    HfP p;
    p.h = (const uint8_t*)d->o; p.w = (const uint8_t*)d->w_packed; p.bo = (const uint8_t*)d->bias; p.res = (const uint8_t*)d->residual;
    p.out = (uint8_t*)d->out; p.rs_out = d->rowstat_out; p.B = d->B; p.N = d->N;
    static const int xm = 8;
    p.xm = xm;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? hs_ff2_launch<APAD_BF16>(p, s) : hs_ff2_launch<APAD_F16>(p, s);
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
    static const int xm_out = 8;
    p.xm = xm_out;
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? hs_out_launch<APAD_BF16>(p, s) : hs_out_launch<APAD_F16>(p, s);
}
