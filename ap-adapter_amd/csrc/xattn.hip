// apad_fused_cross_attention: a whole cross-attention sub-layer of a BasicTransformerBlock in ONE kernel, for key / value
// sets that were hoisted out of the denoise loop -- the adapter's decoupled cross-attention (IPAttnProcessor2_0,
// attention_processor.py:347-470: 8 text keys + La audio keys blended by ap_scale) and the 16-token T5 cross-attention
// (AttnProcessor2_0, :214-294):
//     out = x + to_out( A(q, K1, V1, bias) [+ scale2 * A(q, K2, V2)] ) + b_out,   q = to_q(LayerNorm(x))
// SURVEY 8d's "fused q-proj + attn + blend + out-proj": per sample-forward the launch reads x and writes out once
// (2 x N x C x 2 bytes) instead of six activation passes through three kernels.
//
// Weight-stationary design, C = 256 / 8 heads of 32 (the 1000-token level):
//   * one 512-thread workgroup (8 waves = two per SIMD, so one wave's softmax overlaps the other's MFMAs) owns a tile of
//     128 tokens (4 panels of 32; a panel never straddles two samples)
//   * wave h keeps the 32 rows of Wq of HEAD h in registers (16 A fragments) for the q-projection and the attention of
//     that head, then the 32 rows of Wo of OUTPUT-CHANNEL slice h for the output projection: weights never pass through
//     LDS, and they are read from a fragment-major packing (apad_xattn_pack_weight) so that every wave-load is one
//     contiguous KB
//   * tokens are what moves through LDS: phase 1 copies the tile's RAW rows into the x tile [128][256] and leaves their LayerNorm statistics
//     beside it; phase 2 (wave = head) q_h^T = Wq'_h . x^T, the LayerNorm applied by algebra on the accumulators (Wq' = Wq * gamma, its row
//     sums and Wq . beta come packed: apad_xattn_desc::q_fold) -> S^T = K_h . q_h -> softmax (per segment) -> O_h^T = V_h^T . P^T, the three
//     products chained in registers (each MFMA's C layout is the next one's B operand, the other operand being stored in the matching permuted k
//     order by apad_xattn_pack_kv) -> O tile [128][256]; phase 3 (wave = channel slice) out^T = Wo_slice . O^T + bias, rounded, ADDED IN PLACE to
//     the raw rows the x tile still holds (round 5: x is read from HBM once -- the round-4 kernel re-fetched the residual rows, 1.31 x its
//     algorithmic bytes); phase 4 streams whole rows out
//   * K / V of a (sample, head) are a few KB, fragment-packed at hoist time: coalesced 1 KB loads, L2-resident
#include <type_traits>
#include "rp_shared.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr float XA_LOG2E = 1.4426950408889634f;
constexpr float XA_NEG_BIG = -1.0e30f;

constexpr int XC = 256, XKC = 16, XH = 8, XD = 32;
constexpr int XTM = 128;                  // tokens per workgroup
constexpr int TROWB = XC * 2 + 16;        // tile row stride (bytes): 33 sixteen-byte slots (odd -> conflict-free fragment reads)
constexpr int TILE_BYTES = XTM * TROWB;   // 67 584
constexpr int XA_LDS = 2 * TILE_BYTES + XC * 4 + XTM * 2 * 4 + 2 * XC * 4;  // + bo, per-row (rstd, -mean rstd), the q-projection's fold vectors
constexpr int XMAXSUB = 4;                // <= 128 keys per segment with resident fragments (segment 1: <= 64)
constexpr int XMAXSUB2 = 16;              // <= 512 keys in segment 2 on the chunked form (64-key chunks, running max / sum)
#ifndef XA_SPLIT_Q
#define XA_SPLIT_Q 0
#endif
constexpr bool XA_SPLIT = XA_SPLIT_Q != 0;  // A/B: q-projection of all four panels first (q parked in LDS) vs panel pair by panel pair

struct XaP {
    const uint8_t* x;
    const float* qfold;  // [2][C] or nullptr: LayerNorm folded into the q-projection (apad_xattn_desc::q_fold)
    const uint8_t* wq;   // packed
    const uint8_t* wo;   // packed
    const uint8_t* bo;
    const uint8_t* kv1;  // packed
    const float* bias1;
    const uint8_t* kv2;  // packed
    uint8_t* out;
    int32_t B, N, L1, L2, ppn, npanels, ntiles;
    float eps, scale_log2, scale2;
};

// bytes of one (sample, head) block of a packed K/V set: [K fragments: nsub x 2][V^T fragments: nsub x 2], 1 KB each
__host__ __device__ inline int64_t xa_kv_block(int L) { return (int64_t)((L + 31) / 32) * 4 * 1024; }

// a wave-uniform pointer, pinned to SGPRs: loads through it take the (scalar base + 32-bit lane offset) form instead of a
// 64-bit per-lane address -- the compiler otherwise hoists one such address per weight fragment out of the tile loop (24
// registers per weight), which is what spilled in this 256-register kernel
// (typed as a GLOBAL-address-space pointer: after the integer round trip the compiler no longer infers that, and a generic
//  pointer turns the loads into flat_load, which also ticks lgkmcnt and makes every later wait a vmcnt(0))
typedef const __attribute__((address_space(1))) uint8_t* xa_gptr;
typedef const __attribute__((address_space(1))) u32x4* xa_gptr16;
__device__ __forceinline__ xa_gptr sgpr_ptr(const uint8_t* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (xa_gptr)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ u32x4 xa_ld16(xa_gptr base, uint32_t off) { return *(xa_gptr16)(base + off); }

__device__ __forceinline__ float oct_sum(float v) {
    // 8 consecutive lanes own one token row: two DPP quad permutes + one half-row mirror, no LDS
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    return v;
}

// NS = number of 32-key sub-tiles of the segment (compile time: the fragment registers of an unused second sub-tile would
// push the kernel past the 256-register budget of two waves per SIMD)
template <int DT, int NS> struct XaFrags {
    typename ET<DT>::v8 kf[NS][2];  // [32-key sub-tile][K = 16 step over the head dim]
    typename ET<DT>::v8 vf[2 * NS];  // [K = 16 step over the keys]
};

// coalesced: fragment f of the block is 64 lanes x 16 bytes
template <int DT, int NS> __device__ __forceinline__ void xa_fetch(XaFrags<DT, NS>& f, const uint8_t* blk_, int L, int lane) {
    const xa_gptr blk = sgpr_ptr(blk_);
#pragma unroll
    for (int u = 0; u < NS; ++u) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) f.kf[u][kk] = __builtin_bit_cast(typename ET<DT>::v8, xa_ld16(blk + (u * 2 + kk) * 1024, (uint32_t)(lane * 16)));
    }
    const xa_gptr vb = blk + NS * 2 * 1024;
#pragma unroll
    for (int st = 0; st < 2 * NS; ++st) {
        if (st * 16 >= L) break;
        f.vf[st] = __builtin_bit_cast(typename ET<DT>::v8, xa_ld16(vb + st * 1024, (uint32_t)(lane * 16)));
    }
}

// One softmax segment of the current head: scores from qb (B operand, k = head dim in C-layout order); the probabilities,
// normalised and scaled by `pscale`, are rounded to the storage type and O^T += V^T . P^T.
// Steady-state form: exactly G groups of 8 keys (G and the sub-tile count are compile-time, so the wave's instruction stream
// carries no uniform branches, no masks and no index arithmetic), optional additive key bias (the masked T5 stream).
template <int DT, int G, bool BIAS>
__device__ __forceinline__ void xa_segment_exact(const XaFrags<DT, (G + 3) / 4>& f, const float* bias, float c, float pscale,
                                                 const typename ET<DT>::v8 (&qb)[2], f32x16& o, bool first, int half) {
    using E = ET<DT>;
    constexpr int NS = (G + 3) / 4;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        s[u] = E::mfma32(f.kf[u][0], qb[0], zero16);  // zero accumulator as an inline constant: no register clears
        s[u] = E::mfma32(f.kf[u][1], qb[1], s[u]);
    }
    float tmax = XA_NEG_BIG;
    if (BIAS) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + 8 * g + 4 * half);  // keys 8g + 4 half + j
            const float bj[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = __builtin_fmaf(s[g >> 2][4 * (g & 3) + j], c, bj[j] * XA_LOG2E);
                s[g >> 2][4 * (g & 3) + j] = v;
                tmax = fmaxf(tmax, v);
            }
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) tmax = fmaxf(tmax, s[g >> 2][4 * (g & 3) + j]);
    }
    const float nm = BIAS ? -half_max(tmax) : -half_max(tmax) * c;  // c > 0
    float sum0 = 0.f, sum1 = 0.f;  // two chains: the adds are latency-bound otherwise
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            const int r = 4 * (g & 3) + j;
            const float a0 = s[g >> 2][r], a1 = s[g >> 2][r + 1];
            const float v0 = __builtin_amdgcn_exp2f(BIAS ? a0 + nm : __builtin_fmaf(a0, c, nm));
            const float v1 = __builtin_amdgcn_exp2f(BIAS ? a1 + nm : __builtin_fmaf(a1, c, nm));
            s[g >> 2][r] = v0;
            s[g >> 2][r + 1] = v1;
            sum0 += v0;
            sum1 += v1;
        }
    const float w = pscale * __builtin_amdgcn_rcpf(half_sum(sum0 + sum1));
#pragma unroll
    for (int st = 0; st < (G + 1) / 2; ++st) {
        typename E::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool live = 2 * st + (j >> 2) < G;  // compile time: the second group of an odd last step is all zeros
            pf[j] = live ? (typename E::elem)(s[st >> 1][(st & 1) * 8 + j] * w) : (typename E::elem)0.f;
        }
        // (`first` is a literal at every call site: the first product of a panel takes a constant-zero accumulator)
        o = (first && st == 0) ? E::mfma32(f.vf[0], pf, zero16) : E::mfma32(f.vf[st], pf, o);
    }
}

// General form: run-time length (any L <= 32 NS) and optional bias; groups of 8 keys that lie entirely past the segment are
// skipped (wave-uniform branches), partial groups masked.
template <int DT, int NS>
__device__ __forceinline__ void xa_segment(const XaFrags<DT, NS>& f, int L, const float* bias, float c, float pscale,
                                           const typename ET<DT>::v8 (&qb)[2], f32x16& o, bool first, int half) {
    using E = ET<DT>;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        s[u] = E::mfma32(f.kf[u][0], qb[0], zero16);
        s[u] = E::mfma32(f.kf[u][1], qb[1], s[u]);
    }
    float tmax = XA_NEG_BIG;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (u * 32 + g * 8 >= L) continue;  // uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = g * 4 + j;
                const int key = u * 32 + g * 8 + 4 * half + j;
                float v = s[u][r] * c;
                if (bias) v += bias[key < L ? key : L - 1] * XA_LOG2E;
                v = key < L ? v : XA_NEG_BIG;
                s[u][r] = v;
                tmax = fmaxf(tmax, v);
            }
        }
    }
    tmax = half_max(tmax);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (u * 32 + g * 8 >= L) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = g * 4 + j;
                const float e = __builtin_amdgcn_exp2f(s[u][r] - tmax);
                s[u][r] = e;
                sum += e;
            }
        }
    }
    const float w = pscale * __builtin_amdgcn_rcpf(half_sum(sum));
#pragma unroll
    for (int st = 0; st < 2 * NS; ++st) {
        if (st * 16 >= L) break;
        typename E::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)(s[st >> 1][(st & 1) * 8 + j] * w);  // skipped groups hold 0 (zero K rows)
        o = (first && st == 0) ? E::mfma32(f.vf[0], pf, zero16) : E::mfma32(f.vf[st], pf, o);
    }
}

// One 64-key chunk of a LONG second segment (more than 128 audio tokens: pooling 1 / the mixed poolings of the cfg-3 sweep, AudioMAE.py:148-182):
// scores of the chunk, running maximum m (raw score domain) and per-lane partial sums l0 / l1 of the query's row, the accumulator rescaled when the
// maximum moves, O2^T += V^T . P^T with the UN-normalised probabilities rounded to the storage type (the flash form apad_attention uses for such
// lengths); the caller divides by the sum and applies ap_scale at the end.
template <int DT>
__device__ __forceinline__ void xa_chunk64_scores(f32x16 (&s)[2], const typename ET<DT>::v8 (&kf)[2][2], const typename ET<DT>::v8 (&qb)[2]) {
    using E = ET<DT>;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        s[u] = E::mfma32(kf[u][0], qb[0], zero16);
        s[u] = E::mfma32(kf[u][1], qb[1], s[u]);
    }
}
template <int DT>
__device__ __forceinline__ void xa_chunk64_fold(f32x16 (&s)[2], const typename ET<DT>::v8 (&vf)[4], float c, f32x16& o2, float& m, float& l0, float& l1) {
    using E = ET<DT>;
    float tmax = s[0][0];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[u][r]);
    const float mnew = fmaxf(m, half_max(tmax));
    const float alpha = __builtin_amdgcn_exp2f((m - mnew) * c);  // c > 0; the first chunk: exp2(-huge) = 0 against a zero accumulator
    const float nm = -mnew * c;
    float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float v0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r], c, nm));
            const float v1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][r + 1], c, nm));
            s[u][r] = v0;
            s[u][r + 1] = v1;
            sum0 += v0;
            sum1 += v1;
        }
    // (a wave-uniform "rescale only when some maximum jumped" branch was measured: 93 -> 109 us at 512 keys -- the branch costs the straight-line
    //  schedule more than the 18 multiplies it skips)
    l0 = __builtin_fmaf(l0, alpha, sum0);
    l1 = __builtin_fmaf(l1, alpha, sum1);
#pragma unroll
    for (int r = 0; r < 16; ++r) o2[r] *= alpha;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        typename E::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[st >> 1][(st & 1) * 8 + j];
        o2 = E::mfma32(vf[st], pf, o2);
    }
    m = mnew;
}

// A C-layout accumulator tile (lane = token, registers 4 g + j = channels 8 g + 4 half + j of a 32-channel slice) -> LDS row pieces of 16 bytes:
// the two half-waves exchange 8-byte pieces (v_permlane32_swap) so that the lower half owns channels 0..15 of the token's slice and the upper
// half channels 16..31, each as two whole 16-byte stores (the 8-byte stores this replaces were 2-way bank conflicts on the 33-slot rows and
// twice the store instructions).  `add` = per-register addend (bias) or nullptr; dst = the token's row + the slice's byte offset.
template <int DT, bool RES = false> __device__ __forceinline__ void xa_store_slice(uint8_t* dst, const f32x16& o, const float* add, int half) {
    using E = ET<DT>;
    uint32_t d[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        typename E::v4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = (typename E::elem)(add ? o[4 * g + j] + add[4 * g + j] : o[4 * g + j]);
        const uint2 u = __builtin_bit_cast(uint2, y);
        d[g][0] = u.x;
        d[g][1] = u.y;
    }
    uint4 pc[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const auto a = __builtin_amdgcn_permlane32_swap(d[g][0], d[g + 2][0], false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(d[g][1], d[g + 2][1], false, false);
        pc[g] = make_uint4(a[0], b[0], a[1], b[1]);
    }
    uint8_t* q = dst + half * 32;
    if constexpr (RES) {  // += what the destination holds (the residual), after the rounding above -- the order of the un-fused chain
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            float f[8], r[8];
            unpack8<DT>(pc[g], f);
            unpack8<DT>(*reinterpret_cast<const uint4*>(q + g * 16), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += r[e];
            pc[g] = pack8<DT>(f);
        }
    }
    *reinterpret_cast<uint4*>(q) = pc[0];
    *reinterpret_cast<uint4*>(q + 16) = pc[1];
}

#ifdef XATTN_TRACE  // probe build (tools/xattn_trace.py): s_memtime stamps of wave 0 / wave 7 at the phase boundaries
__device__ unsigned long long g_xa_trace[1024 * 32];
#define XA_STAMP(i) \
    do { if (lane == 0 && (wave == 0 || wave == 7)) g_xa_trace[(tile * 2 + (wave == 7)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XA_STAMP(i)
#endif

// NS1 / NS2 = 32-key sub-tiles per segment (NS2 = 0: single segment).  G1 > 0 selects the exact form: L1 = 8 G1 and L2 = 8 G2
// keys precisely, BIAS1 = segment 1 carries a key bias; G1 = 0: lengths and bias are run-time (any L <= 32 NS).
// CHUNK: segment 2 is LONG (L2 = 64 * n <= 512 keys): its fragments are fetched 64 keys at a time (NS2 = 2 is the chunk's shape) and folded in with a
// running max / sum (xa_chunk64); segment 1 is the exact 8-key text segment.
template <int DT, int NS1, int NS2, int G1 = 0, int G2 = 0, bool BIAS1 = false, bool CHUNK = false>
__global__ __launch_bounds__(512) void xattn_kernel(XaP p) {
    constexpr bool DUAL = NS2 > 0;
    constexpr bool EXACT = G1 > 0;
    // more than 64 audio keys (La = 128: the timbre / accompaniment presets, config.py:8-11, :50-53): the segment's 16 K / V^T
    // fragments + four score tiles do not fit beside the stationary Wq rows, so the q-projection of all four panels runs first (q parked
    // in LDS, Wq registers dead afterwards), the fragments are fetched behind it, and panels are attended one at a time
    constexpr bool BIG2 = NS2 > 2;
    constexpr bool SPLIT = XA_SPLIT || BIG2 || CHUNK;
    using E = ET<DT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const xt = smem;                 // x^ tile, later the output tile
    uint8_t* const ot = smem + TILE_BYTES;    // O tile
    float* const lbo = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
    float* const lst = lbo + XC;           // [128][2]: rstd, -mean * rstd of the tile's rows
    float* const lfold = lst + XTM * 2;    // [2][256]: row sums of the folded Wq, Wq . beta
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: per-wave bases stay in SGPRs (registers are tight)
    const int half = lane >> 5, l31 = lane & 31;
    const int oct = lane & 7;
    // Persistent workgroups (one per CU: the two LDS tiles fill it), each walking virtual ids v = blockIdx.x, + gridDim.x, ...
    // XCD-aware order: workgroup id w runs on XCD w % 8 (observed, speed only) and gridDim.x is a multiple of 8, so v % 8 is
    // that XCD; each XCD walks a contiguous range of tiles, i.e. whole samples, whose packed K/V are then fetched into one L2.
    const int per = (p.ntiles + 7) >> 3;
    auto tile_of = [&](int v) { return (v & 7) * per + (v >> 3); };
    auto next_valid = [&](int v) {  // next virtual id of this workgroup that maps to an existing tile, or -1
        for (v += gridDim.x; v < 8 * per; v += gridDim.x)
            if (tile_of(v) < p.ntiles) return v;
        return -1;
    };
    // rows of a tile owned by this thread in phases 1 / 4: a wave owns 16 rows; 8 consecutive lanes own one row and move it as
    // interleaved 16-byte chunks (chunk = 8 i + lane % 8), so every load / store instruction of the wave covers 8 rows x 128
    // contiguous bytes = whole cache lines
    int trow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) trow[j] = wave * 16 + j * 8 + (lane >> 3);
    // (addresses are a uniform base + a 32-bit per-lane byte offset -- B * N * 512 < 4 GB is checked on the host -- so that the
    //  compiler keeps bases in SGPRs: hoisted 64-bit per-lane pointers were what spilled in this 256-register kernel)
    constexpr uint32_t NOROW = 0xffffffffu;
    auto rows_of = [&](int tile, uint32_t (&xoff)[2]) {  // byte offset of the thread's chunk 0 in its two rows, or NOROW
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int panel = tile * 4 + (trow[j] >> 5);
            const int b = panel / p.ppn, q = (panel - b * p.ppn) * 32 + (trow[j] & 31);
            xoff[j] = (panel < p.npanels && q < p.N) ? (uint32_t)(b * p.N + q) * (XC * 2) + oct * 16 : NOROW;
        }
    };
    auto load_rows = [&](uint4 (&r)[2][4], const uint32_t (&xoff)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t o = xoff[j] != NOROW ? xoff[j] : oct * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[j][i] = *reinterpret_cast<const uint4*>(p.x + (o + i * 128));
        }
    };

    int v = blockIdx.x;
    if (tile_of(v) >= p.ntiles) v = next_valid(v);
    if (v < 0) return;
    for (int i = tid; i < XC; i += 512) lbo[i] = p.bo ? ld_elem<DT>(p.bo, i) : 0.f;
    if (p.qfold != nullptr) lfold[tid] = p.qfold[tid];  // 512 threads = 2 x 256 floats
    uint32_t xoff[2];
    uint4 raw[2][4];  // the tile's un-normalised rows: requested one tile ahead (under phase 3 of the previous tile)
    rows_of(tile_of(v), xoff);
    load_rows(raw, xoff);

    while (true) {
    const int tile = tile_of(v);
    // next tile and its row offsets, computed here: the stretch between the Wo loads and their first use below must stay
    // straight-line code (loops / exec-masked branches there make the compiler's wait-count pass fall back to vmcnt(0), i.e.
    // wait for the NEXT tile's rows before this tile's output projection)
    const int vnext = next_valid(v);
    uint32_t xoff_next[2];
    rows_of(tile_of(vnext >= 0 ? vnext : v), xoff_next);
    if (vnext < 0) xoff_next[0] = xoff_next[1] = NOROW;  // the last tile: the loads stay unconditional but all hit row 0 (round 4 re-requested the
                                                        // tile's own rows here: half of all tiles fetched twice, 12 MB of the launch's extra reads)
    XA_STAMP(0);
#ifdef XATTN_TRACE
    if (lane == 0 && wave == 0) g_xa_trace[tile * 32 + 12] = wall_clock64();
#endif
    // ---- phase 1: the tile's RAW rows -> x tile (they stay there: B operand of the q-projection, then the residual, added in place in phase
    //      3); with a LayerNorm: the rows' statistics -> lst (the normalisation itself is algebra on the q accumulators) ----
    {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(xt + trow[j] * TROWB + oct * 16 + i * 128) = raw[j][i];
            if (p.qfold != nullptr) {
                float f[32];  // unpacked once; two-pass statistics from registers
#pragma unroll
                for (int i = 0; i < 4; ++i) unpack8<DT>(raw[j][i], f + 8 * i);
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int e = 0; e < 32; e += 2) {
                    s0 += f[e];
                    s1 += f[e + 1];
                }
                const float mean = oct_sum(s0 + s1) * (1.0f / XC);
                float q0 = 0.f, q1 = 0.f;
#pragma unroll
                for (int e = 0; e < 32; e += 2) {
                    const float d0 = f[e] - mean, d1 = f[e + 1] - mean;
                    q0 = __builtin_fmaf(d0, d0, q0);
                    q1 = __builtin_fmaf(d1, d1, q1);
                }
                const float rstd = rsqrtf(oct_sum(q0 + q1) * (1.0f / XC) + p.eps);
                if (oct == 0) *reinterpret_cast<float2*>(lst + trow[j] * 2) = make_float2(rstd, -mean * rstd);
            }
        }
    }
    // ---- weights of this wave: head `wave` of Wq (A operand of the q-projection); requested here, behind the LayerNorm's
    //      temporaries, so that its 64 registers are not live (and spilled) during phase 1; the barrier hides the latency ----
    typename E::v8 wf[XKC];
    {
        const xa_gptr wp = sgpr_ptr(p.wq + wave * (XKC * 1024));
#pragma unroll
        for (int kk = 0; kk < XKC; ++kk) wf[kk] = __builtin_bit_cast(typename ET<DT>::v8, xa_ld16(wp + kk * 1024, (uint32_t)(lane * 16)));
    }
    XA_STAMP(1);
    __syncthreads();
    XA_STAMP(2);

    // ---- phase 2: wave = head h.  (a) q-projection of all four panels at once: q_h^T [32 dims x 32 tokens] = Wq_h . x^^T, A =
    //      stationary registers, B = token fragments from the x^ tile; four independent accumulator chains keep the matrix
    //      pipe issuing back to back.  q (storage type, already in the B-operand register order of the score product) is
    //      parked in this wave's own 64-byte-per-token slots of the O tile.  (b) attention, two panels interleaved (two
    //      independent MFMA -> softmax -> MFMA chains per wave: a single chain is latency-bound and two waves per SIMD do not
    //      hide it), O_h over the parked q.  No barrier between (a) and (b): the slots are private to the wave. ----
    {
        const int h = wave;
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // LayerNorm by algebra on a q accumulator tile (lane = token l31 of panel pn, register r = head dim (r & 3) + 8 (r >> 2) + 4 half):
        // q = rstd * (W' x - mean * rowsum(W')) + W . beta  =  acc * rstd + ((-mean rstd) * cs[dim] + bb[dim]); the fold vectors come from LDS
        // right here (short-lived: the registers of this phase are spoken for)
        auto fold_q = [&](f32x16& acc, int pn) {
            if (p.qfold == nullptr) return;  // (wave-uniform)
            const float2 st = *reinterpret_cast<const float2*>(lst + (pn * 32 + l31) * 2);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 cs = *reinterpret_cast<const float4*>(lfold + h * XD + 8 * g + 4 * half);
                const float4 bb = *reinterpret_cast<const float4*>(lfold + XC + h * XD + 8 * g + 4 * half);
                acc[4 * g + 0] = __builtin_fmaf(acc[4 * g + 0], st.x, __builtin_fmaf(st.y, cs.x, bb.x));
                acc[4 * g + 1] = __builtin_fmaf(acc[4 * g + 1], st.x, __builtin_fmaf(st.y, cs.y, bb.y));
                acc[4 * g + 2] = __builtin_fmaf(acc[4 * g + 2], st.x, __builtin_fmaf(st.y, cs.z, bb.z));
                acc[4 * g + 3] = __builtin_fmaf(acc[4 * g + 3], st.x, __builtin_fmaf(st.y, cs.w, bb.w));
            }
        };
        // K / V fragments of this head for the tile's first sample: requested here, so that their L2 latency hides under the
        // q-projection; the panels of a tile share them unless the tile crosses a sample boundary
        XaFrags<DT, NS1> f1;
        XaFrags<DT, DUAL ? NS2 : 1> f2;
        auto fetch_kv = [&](int b) {
            xa_fetch<DT, NS1>(f1, p.kv1 + ((int64_t)b * XH + h) * xa_kv_block(p.L1), p.L1, lane);
            if (DUAL) xa_fetch<DT, DUAL ? NS2 : 1>(f2, p.kv2 + ((int64_t)b * XH + h) * xa_kv_block(p.L2), p.L2, lane);
        };
        const int bfirst = (tile * 4) / p.ppn;
        if constexpr (EXACT && !BIG2 && !CHUNK) fetch_kv(bfirst < p.B ? bfirst : p.B - 1);  // (general form: after the q-projection -- registers)
        if constexpr (SPLIT) {
            f32x16 qa[4];
            const uint8_t* bt = xt + l31 * TROWB + half * 16;
            typename E::v8 fb[2][4];
            auto ldk = [&](int buf, int kk) {
#pragma unroll
                for (int pn = 0; pn < 4; ++pn) fb[buf][pn] = as_v8<DT>(*reinterpret_cast<const uint4*>(bt + pn * 32 * TROWB + kk * 32));
            };
            ldk(0, 0);
#pragma unroll
            for (int kk = 0; kk < XKC; ++kk) {
                const int cur = kk & 1;
                if (kk + 1 < XKC) ldk(cur ^ 1, kk + 1);
#pragma unroll
                for (int pn = 0; pn < 4; ++pn) {
                    asm volatile("" : "+v"(fb[cur][pn]) : : "memory");  // wait for exactly this k-step's reads
                    qa[pn] = E::mfma32(wf[kk], fb[cur][pn], kk == 0 ? zero16 : qa[pn]);
                }
            }
            // the reference materialises q in the storage type; registers 0..7 / 8..15 are the two K = 16 steps over the head dim
#pragma unroll
            for (int pn = 0; pn < 4; ++pn) {
                typename E::v8 qb[2];
                fold_q(qa[pn], pn);
#pragma unroll
                for (int r = 0; r < 16; ++r) qb[r >> 3][r & 7] = (typename E::elem)qa[pn][r];
                uint8_t* qd = ot + (pn * 32 + l31) * TROWB + h * 64 + half * 32;
                *reinterpret_cast<uint4*>(qd) = as_u4<DT>(qb[0]);
                *reinterpret_cast<uint4*>(qd + 16) = as_u4<DT>(qb[1]);
            }
        }
        if constexpr ((!EXACT || BIG2) && !CHUNK) fetch_kv(bfirst < p.B ? bfirst : p.B - 1);
        XA_STAMP(3);
        // attention of `n` panels (pp, pp + 1) of ONE sample, interleaved: the panels share the K / V fragments
        auto attend = [&](int pp, auto n_tag, int b) {
            constexpr int NP = decltype(n_tag)::value;
            typename E::v8 qb[NP][2];
            if constexpr (SPLIT) {
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const uint8_t* qd = ot + ((pp + u) * 32 + l31) * TROWB + h * 64 + half * 32;
                    qb[u][0] = as_v8<DT>(*reinterpret_cast<const uint4*>(qd));
                    qb[u][1] = as_v8<DT>(*reinterpret_cast<const uint4*>(qd + 16));
                }
            } else {
                // q-projection of these panels right here: NP accumulator chains, fragment reads one k-step ahead
                f32x16 qa[NP];
                const uint8_t* bt = xt + (pp * 32 + l31) * TROWB + half * 16;
                typename E::v8 fb[2][NP];
                auto ldk = [&](int buf, int kk) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) fb[buf][u] = as_v8<DT>(*reinterpret_cast<const uint4*>(bt + u * 32 * TROWB + kk * 32));
                };
                ldk(0, 0);
#pragma unroll
                for (int kk = 0; kk < XKC; ++kk) {
                    const int cur = kk & 1;
                    if (kk + 1 < XKC) ldk(cur ^ 1, kk + 1);
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        asm volatile("" : "+v"(fb[cur][u]) : : "memory");
                        qa[u] = E::mfma32(wf[kk], fb[cur][u], kk == 0 ? zero16 : qa[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    fold_q(qa[u], pp + u);
#pragma unroll
                    for (int r = 0; r < 16; ++r) qb[u][r >> 3][r & 7] = (typename E::elem)qa[u][r];
                }
            }
            const float* bias1 = p.bias1 ? p.bias1 + (int64_t)b * p.L1 : nullptr;
            f32x16 o[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                o[u] = zero16;
                // probabilities are normalised (and the audio segment scaled by ap_scale, attention_processor.py:454) BEFORE they
                // are rounded for the P.V product -- as the reference's softmax output is -- so both segments accumulate
                // into ONE O accumulator
                if constexpr (EXACT) xa_segment_exact<DT, G1, BIAS1>(f1, bias1, p.scale_log2, 1.0f, qb[u], o[u], true, half);
                else xa_segment<DT, NS1>(f1, p.L1, bias1, p.scale_log2, 1.0f, qb[u], o[u], true, half);
                if (DUAL) {
                    if constexpr (EXACT && DUAL) xa_segment_exact<DT, (G2 > 0 ? G2 : 1), false>(f2, nullptr, p.scale_log2, p.scale2, qb[u], o[u], false, half);
                    else xa_segment<DT, DUAL ? NS2 : 1>(f2, p.L2, nullptr, p.scale_log2, p.scale2, qb[u], o[u], false, half);
                }
            }
            // O_h^T (lane = token, registers = head dims 8g + 4 half + j) -> O tile [token][h*32 + dim] (over the parked q)
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                xa_store_slice<DT>(ot + ((pp + u) * 32 + l31) * TROWB + h * XD * 2, o[u], nullptr, half);
            }
        };
        // CHUNK form: the panels pp .. pp + CNT - 1 of ONE sample; chunks outer (each chunk's 8 KB of fragments fetched once for all of them), panels inner
        auto chunk_group = [&](int pp, auto cnt_tag, int b) {
            constexpr int CNT = decltype(cnt_tag)::value;
            xa_fetch<DT, NS1>(f1, p.kv1 + ((int64_t)b * XH + h) * xa_kv_block(p.L1), p.L1, lane);
            const int nsub2 = p.L2 >> 5;
            const xa_gptr kb = sgpr_ptr(p.kv2 + ((int64_t)b * XH + h) * xa_kv_block(p.L2));
            const xa_gptr vb = kb + nsub2 * 2048;
            f32x16 o2[CNT];
            float mx[CNT], l0[CNT], l1[CNT];
#pragma unroll
            for (int u = 0; u < CNT; ++u) {
                o2[u] = zero16;
                mx[u] = XA_NEG_BIG;
                l0[u] = l1[u] = 0.f;
            }
            auto load_q = [&](int u, typename E::v8 (&qb)[2]) {
                const uint8_t* qd = ot + ((pp + u) * 32 + l31) * TROWB + h * 64 + half * 32;
                qb[0] = as_v8<DT>(*reinterpret_cast<const uint4*>(qd));
                qb[1] = as_v8<DT>(*reinterpret_cast<const uint4*>(qd + 16));
            };
            // ONE fragment set, refilled in place: the next chunk's K fragments are requested as soon as the last panel's score MFMAs have read the
            // current ones (their latency under that panel's softmax + P.V), its V^T fragments behind the last P.V (under the next chunk's first scores)
            typename E::v8 kf[2][2], vf[4];
            const int nch = nsub2 >> 1;
            auto load_k = [&](int c) {
                c = c < nch ? c : nch - 1;  // (past the end: a harmless re-load keeps the loads unconditional)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        kf[u][kk] = __builtin_bit_cast(typename E::v8, xa_ld16(kb + (c * 4 + u * 2 + kk) * 1024, (uint32_t)(lane * 16)));
            };
            auto load_v = [&](int c) {
                c = c < nch ? c : nch - 1;
#pragma unroll
                for (int st = 0; st < 4; ++st) vf[st] = __builtin_bit_cast(typename E::v8, xa_ld16(vb + (c * 4 + st) * 1024, (uint32_t)(lane * 16)));
            };
            load_k(0);
            load_v(0);
#pragma unroll 1
            for (int c = 0; c < nch; ++c) {
#pragma unroll
                for (int u = 0; u < CNT; ++u) {
                    typename E::v8 qb[2];
                    load_q(u, qb);
                    f32x16 sc[2];
                    xa_chunk64_scores<DT>(sc, kf, qb);
                    if (u == CNT - 1) load_k(c + 1);
                    xa_chunk64_fold<DT>(sc, vf, p.scale_log2, o2[u], mx[u], l0[u], l1[u]);
                    if (u == CNT - 1) load_v(c + 1);
                }
            }
#pragma unroll
            for (int u = 0; u < CNT; ++u) {
                typename E::v8 qb[2];
                load_q(u, qb);
                const float w2 = p.scale2 * __builtin_amdgcn_rcpf(half_sum(l0[u] + l1[u]));  // ap_scale / row sum (attention_processor.py:454)
#pragma unroll
                for (int r = 0; r < 16; ++r) o2[u][r] *= w2;
                // the text segment on top (its probabilities normalised before they are rounded, as in the resident-fragment forms)
                xa_segment_exact<DT, (G1 > 0 ? G1 : 1), false>(f1, nullptr, p.scale_log2, 1.0f, qb, o2[u], false, half);
                xa_store_slice<DT>(ot + ((pp + u) * 32 + l31) * TROWB + h * XD * 2, o2[u], nullptr, half);
            }
        };
        using One = std::integral_constant<int, 1>;
        using Two = std::integral_constant<int, (EXACT && !BIG2) ? 2 : 1>;  // (the general form's / a big segment's fragment sets leave no room for two chains)
        if constexpr (CHUNK) {
            const int blast = (tile * 4 + 3) / p.ppn;
            if (blast == bfirst) {  // wave-uniform; every tile when the sample's panel count is a multiple of 4 (1000 tokens: 32 panels)
                chunk_group(0, std::integral_constant<int, 4>{}, bfirst < p.B ? bfirst : p.B - 1);
            } else {
#pragma unroll 1
                for (int pp = 0; pp < 4; ++pp) {
                    const int bp = (tile * 4 + pp) / p.ppn;
                    chunk_group(pp, One{}, bp < p.B ? bp : p.B - 1);
                }
            }
            XA_STAMP(4);
            XA_STAMP(5);
        } else {
        int bnext = bfirst, rem = tile * 4 - bfirst * p.ppn;  // sample / panel-in-sample walk over the tile's panels
        int bcur = bfirst < p.B ? bfirst : p.B - 1;           // sample whose fragments are loaded
#pragma unroll 1
        for (int pp = 0; pp < 4; pp += 2) {
            int bs[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bs[u] = bnext < p.B ? bnext : p.B - 1;  // tail tile: recompute the last sample's rows (never stored)
                if (++rem == p.ppn) {
                    rem = 0;
                    ++bnext;
                }
            }
            if (Two::value == 2 && bs[0] == bcur && bs[1] == bcur) {  // wave-uniform; 7 tiles of 8 at N = 1000
                attend(pp, Two{}, bcur);
            } else {
#pragma unroll 1
                for (int u = 0; u < 2; ++u) {
                    if (bs[u] != bcur) {  // the tile crosses into the next sample: its fragments (latency exposed, once)
                        bcur = bs[u];
                        fetch_kv(bcur);
                    }
                    attend(pp + u, One{}, bcur);
                }
            }
            XA_STAMP(4 + (pp >> 1));
        }
        }
    }
    // ---- weights of this wave for phase 3: rows (output channels) wave*32.. of Wo ----
    {
        const xa_gptr wp = sgpr_ptr(p.wo + wave * (XKC * 1024));
#pragma unroll
        for (int kk = 0; kk < XKC; ++kk) wf[kk] = __builtin_bit_cast(typename ET<DT>::v8, xa_ld16(wp + kk * 1024, (uint32_t)(lane * 16)));
    }
    XA_STAMP(7);
    __syncthreads();  // O tile complete; the x^ tile is free
    XA_STAMP(8);

    // the NEXT tile's rows are requested now: their HBM latency hides under the output projection instead of standing at the head of the
    // next tile (the residual of THIS tile is in the x tile: nothing is read twice)
    load_rows(raw, xoff_next);

    // ---- phase 3: wave = output-channel slice.  out^T [32 channels x 32 tokens] = Wo_slice . O^T + bias, rounded, + the raw x rows the x tile
    //      still holds (the (token, channel slice) block a lane pair reads is the block it writes: in place) ----
    {
#pragma unroll  // (a rolled loop makes the wait-count pass emit vmcnt(0) at its header: see the prefetch above)
        for (int pn = 0; pn < 4; pn += 2) {
            const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            f32x16 ya = zero16, yb = zero16;  // two panels at once: two independent accumulator chains
            const uint8_t* b0 = ot + (pn * 32 + l31) * TROWB + half * 16;
            const uint8_t* b1 = b0 + 32 * TROWB;
            // fragment reads double-buffered in groups of two k-steps per panel (registers: the next tile's rows and this
            // tile's residual rows are in flight through this phase)
            typename E::v8 fa[2][2], ga[2][2];
            auto ldg = [&](int buf, int g) {
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    fa[buf][cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(b0 + (g * 2 + cc) * 32));
                    ga[buf][cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(b1 + (g * 2 + cc) * 32));
                }
            };
            ldg(0, 0);
#pragma unroll
            for (int g = 0; g < XKC / 2; ++g) {
                const int cur = g & 1;
                if (g + 1 < XKC / 2) ldg(cur ^ 1, g + 1);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    asm volatile("" : "+v"(fa[cur][cc]), "+v"(ga[cur][cc]) : : "memory");  // wait for exactly this group's reads
                    ya = E::mfma32(wf[g * 2 + cc], fa[cur][cc], ya);
                    yb = E::mfma32(wf[g * 2 + cc], ga[cur][cc], yb);
                }
            }
            uint8_t* d0 = xt + (pn * 32 + l31) * TROWB + wave * 32 * 2;
            float bo[16];  // (short-lived: registers are tight here)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(lbo + wave * 32 + 8 * g + 4 * half);
                bo[4 * g] = b4.x; bo[4 * g + 1] = b4.y; bo[4 * g + 2] = b4.z; bo[4 * g + 3] = b4.w;
            }
            xa_store_slice<DT, true>(d0, ya, bo, half);
            xa_store_slice<DT, true>(d0 + 32 * TROWB, yb, bo, half);
        }
    }
    XA_STAMP(9);
    __syncthreads();
    XA_STAMP(10);

    // ---- phase 4: the finished rows stream out, whole cache lines per store instruction ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (xoff[j] == NOROW) continue;
        const uint8_t* src = xt + trow[j] * TROWB + oct * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(p.out + (xoff[j] + i * 128)) = *reinterpret_cast<const uint4*>(src + i * 128);
    }
    XA_STAMP(11);
#ifdef XATTN_TRACE
    if (lane == 0 && wave == 0) {
        g_xa_trace[tile * 32 + 13] = wall_clock64();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_xa_trace[tile * 32 + 14] = xcc & 0xf;
    }
#endif
    if (vnext < 0) break;
    v = vnext;
    xoff[0] = xoff_next[0];
    xoff[1] = xoff_next[1];
    __syncthreads();  // phase 4 has read the output tile before the next tile's phase 1 overwrites it
    }
}

// ---- packing kernels (one-off per weight / per hoisted K,V set) ----
// W [256][256] -> [slice s = row / 32][kk][lane][8] = W[s*32 + l31][kk*16 + half*8 + e]
template <int DT> __global__ void xa_pack_w_kernel(const uint8_t* w, uint8_t* out, int64_t ldw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // one 16-byte chunk each: 8 slices x 16 kk x 64 lanes
    if (idx >= XH * XKC * 64) return;
    const int lane = idx & 63, kk = (idx >> 6) & 15, s = idx >> 10;
    const int l31 = lane & 31, half = lane >> 5;
    *reinterpret_cast<uint4*>(out + (int64_t)idx * 16) =
        *reinterpret_cast<const uint4*>(w + ((int64_t)(s * 32 + l31) * ldw + kk * 16 + half * 8) * 2);
}

// k [B][L][256], vt [B][8][32][Lpad] -> per (b, h): K fragments [sub][kk][lane][8] = K[b][sub*32 + l31][h*32 + kk*16 + perm(half, e)]
// (rows >= L: 0), V^T fragments [st][lane][8] = vt[b][h][l31][st*16 + perm(half, e)], perm(half, e) = 4 half + e (e < 4),
// 8 + 4 half + e - 4 (e >= 4): the key / dim order in which an MFMA C layout holds them (see xattn_kernel)
template <int DT> __global__ void xa_pack_kv_kernel(const uint8_t* k, const uint8_t* vt, uint8_t* out, int B, int L, int Lpad,
                                                    int64_t k_sb, int64_t k_sl, int64_t vt_sb) {
    using elem = typename ET<DT>::elem;
    const int nsub = (L + 31) >> 5;
    const int bh = blockIdx.x, b = bh / XH, h = bh % XH;
    uint8_t* blk = out + (int64_t)bh * xa_kv_block(L);
    const elem* kb = reinterpret_cast<const elem*>(k) + (int64_t)b * k_sb + h * XD;
    const elem* vb = reinterpret_cast<const elem*>(vt) + (int64_t)b * vt_sb + (int64_t)h * XD * Lpad;
    for (int idx = threadIdx.x; idx < nsub * 4 * 64; idx += blockDim.x) {
        const int lane = idx & 63, fr = idx >> 6;
        const int l31 = lane & 31, half = lane >> 5;
        elem v[8];
        if (fr < nsub * 2) {
            const int sub = fr >> 1, kk = fr & 1, key = sub * 32 + l31;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int d = kk * 16 + (e < 4 ? 4 * half + e : 8 + 4 * half + e - 4);
                v[e] = key < L ? kb[(int64_t)key * k_sl + d] : (elem)0.f;
            }
        } else {
            const int st = fr - nsub * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int key = st * 16 + (e < 4 ? 4 * half + e : 8 + 4 * half + e - 4);
                v[e] = key < L ? vb[(int64_t)l31 * Lpad + key] : (elem)0.f;
            }
        }
        *reinterpret_cast<uint4*>(blk + (int64_t)idx * 16) = *reinterpret_cast<const uint4*>(v);
    }
}

template <int DT, int NS1, int NS2, int G1 = 0, int G2 = 0, bool BIAS1 = false, bool CHUNK = false> int xa_launch(const XaP& p, hipStream_t s) {
    auto kern = xattn_kernel<DT, NS1, NS2, G1, G2, BIAS1, CHUNK>;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), XA_LDS, &devs) != 0) return -1;
    // persistent: one workgroup per CU (a multiple of 8 so that virtual id % 8 stays the XCD), fewer when there are fewer tiles
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        return n / 8 * 8;
    }();
    const int want = ((p.ntiles + 7) / 8) * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)(want < ncu ? want : ncu)), dim3(512), XA_LDS, s, p);
    return apad_check_launch("apad_fused_cross_attention");
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

#ifdef XATTN_TRACE
extern "C" int apad_xattn_trace_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_xa_trace), (size_t)n * sizeof(unsigned long long));
}
#endif

extern "C" int64_t apad_xattn_packed_kv_bytes(int32_t B, int32_t L) { return (int64_t)B * XH * xa_kv_block(L); }

extern "C" int apad_xattn_pack_weight(const void* w, void* packed, int64_t ldw, int32_t dtype, void* stream) {
    APAD_CHECK(w && packed && al16(w) && al16(packed) && ldw % 8 == 0 && ldw >= XC, "apad_xattn_pack_weight: bad operand");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_xattn_pack_weight: dtype %d not supported", dtype);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_BF16)
        hipLaunchKernelGGL((xa_pack_w_kernel<APAD_BF16>), dim3(XH * XKC * 64 / 256), dim3(256), 0, s, (const uint8_t*)w, (uint8_t*)packed, ldw);
    else
        hipLaunchKernelGGL((xa_pack_w_kernel<APAD_F16>), dim3(XH * XKC * 64 / 256), dim3(256), 0, s, (const uint8_t*)w, (uint8_t*)packed, ldw);
    return apad_check_launch("apad_xattn_pack_weight");
}

extern "C" int apad_xattn_pack_kv(const void* k, const void* vt, void* packed, int32_t B, int32_t L, int32_t Lpad, int64_t k_stride_b,
                                  int64_t k_stride_l, int64_t vt_stride_b, int32_t dtype, void* stream) {
    APAD_CHECK(k && vt && packed && al16(packed), "apad_xattn_pack_kv: bad operand");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_xattn_pack_kv: dtype %d not supported", dtype);
    APAD_CHECK(B > 0 && L > 0 && L <= 32 * XMAXSUB2 && Lpad >= L && Lpad % 16 == 0 && Lpad >= ((L + 15) / 16) * 16,
               "apad_xattn_pack_kv: need 0 < L <= %d and Lpad >= L (B=%d L=%d Lpad=%d)", 32 * XMAXSUB2, B, L, Lpad);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_BF16)
        hipLaunchKernelGGL((xa_pack_kv_kernel<APAD_BF16>), dim3((unsigned)(B * XH)), dim3(256), 0, s, (const uint8_t*)k, (const uint8_t*)vt,
                           (uint8_t*)packed, B, L, Lpad, k_stride_b, k_stride_l, vt_stride_b);
    else
        hipLaunchKernelGGL((xa_pack_kv_kernel<APAD_F16>), dim3((unsigned)(B * XH)), dim3(256), 0, s, (const uint8_t*)k, (const uint8_t*)vt,
                           (uint8_t*)packed, B, L, Lpad, k_stride_b, k_stride_l, vt_stride_b);
    return apad_check_launch("apad_xattn_pack_kv");
}

extern "C" int apad_fused_cross_attention(const apad_xattn_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_fused_cross_attention: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_fused_cross_attention: dtype %d not supported", d->dtype);
    const bool long2 = d->L2 > 32 * XMAXSUB;  // the chunked form: 8 text keys + 64 n <= 512 audio keys, no key bias
    if (d->C != XC || d->heads != XH || d->L1 > 64 || d->L2 > 32 * XMAXSUB2 || (long2 && (d->L1 != 8 || d->L2 % 64 != 0 || d->key_bias != nullptr))) {
        apad_set_error("apad_fused_cross_attention: C=%d heads=%d L1=%d L2=%d outside the kernel envelope (C 256, 8 heads, <= 64 + %d keys, or 8 + 64 n <= %d "
                       "unmasked)", d->C, d->heads, d->L1, d->L2, 32 * XMAXSUB, 32 * XMAXSUB2);
        return -3;
    }
    APAD_CHECK(d->x && d->wq_packed && d->wo_packed && d->kv1_packed && d->out, "apad_fused_cross_attention: null operand");
    APAD_CHECK(d->B > 0 && d->N > 0 && d->L1 > 0, "apad_fused_cross_attention: bad geometry B=%d N=%d L1=%d", d->B, d->N, d->L1);
    const bool dual = d->L2 > 0;
    if (dual) APAD_CHECK(d->kv2_packed != nullptr, "apad_fused_cross_attention: segment 2 needs kv2_packed");
    APAD_CHECK((d->ln_gamma == nullptr) == (d->ln_beta == nullptr), "apad_fused_cross_attention: LayerNorm needs gamma and beta");
    APAD_CHECK((d->ln_gamma == nullptr) == (d->q_fold == nullptr),
               "apad_fused_cross_attention: a LayerNorm is applied by algebra (ABI 8): pass wq_packed of round(W * gamma) and q_fold (see the header)");
    APAD_CHECK(al16(d->x) && al16(d->wq_packed) && al16(d->wo_packed) && al16(d->kv1_packed) && al16(d->out) && al16(d->kv2_packed) &&
                   al16(d->q_fold),
               "apad_fused_cross_attention: pointers must be 16-byte aligned");
    XaP p;
    p.x = (const uint8_t*)d->x; p.qfold = d->q_fold;
    p.wq = (const uint8_t*)d->wq_packed; p.wo = (const uint8_t*)d->wo_packed; p.bo = (const uint8_t*)d->bo;
    p.kv1 = (const uint8_t*)d->kv1_packed; p.bias1 = d->key_bias; p.kv2 = (const uint8_t*)d->kv2_packed; p.out = (uint8_t*)d->out;
    p.B = d->B; p.N = d->N; p.L1 = d->L1; p.L2 = d->L2;
    p.ppn = (d->N + 31) / 32;
    p.npanels = d->B * p.ppn;
    p.ntiles = (p.npanels + 3) / 4;
    p.eps = d->ln_eps; p.scale_log2 = d->softmax_scale * XA_LOG2E; p.scale2 = d->scale2;
    hipStream_t s = (hipStream_t)stream;
    const int ns1 = (d->L1 + 31) / 32, ns2 = (d->L2 + 31) / 32;
    if (long2) return d->dtype == APAD_BF16 ? xa_launch<APAD_BF16, 1, 2, 1, 8, false, true>(p, s) : xa_launch<APAD_F16, 1, 2, 1, 8, false, true>(p, s);
    // exact forms: the adapter's presets (8 text tokens + 8 / 32 / 64 audio tokens, no mask) and the masked 16-token T5 stream
#define XA_EXACT(g1, g2, hasb)                                                                                          \
    if (d->L1 == 8 * g1 && d->L2 == 8 * g2 && (d->key_bias != nullptr) == hasb)                                        \
        return d->dtype == APAD_BF16 ? xa_launch<APAD_BF16, (g1 + 3) / 4, (g2 + 3) / 4, g1, g2, hasb>(p, s)              \
                                     : xa_launch<APAD_F16, (g1 + 3) / 4, (g2 + 3) / 4, g1, g2, hasb>(p, s);
    XA_EXACT(1, 4, false) XA_EXACT(1, 1, false) XA_EXACT(1, 8, false) XA_EXACT(2, 0, true) XA_EXACT(1, 16, false)
#undef XA_EXACT
#define XA_CASE(A, B2) \
    if (ns1 == A && ns2 == B2) return d->dtype == APAD_BF16 ? xa_launch<APAD_BF16, A, B2>(p, s) : xa_launch<APAD_F16, A, B2>(p, s);
    XA_CASE(1, 0) XA_CASE(2, 0) XA_CASE(1, 1) XA_CASE(1, 2) XA_CASE(2, 1) XA_CASE(2, 2)
#undef XA_CASE
    apad_set_error("apad_fused_cross_attention: no kernel for L1=%d L2=%d", d->L1, d->L2);
    return -1;
}
