// apad_attention: softmax(Q K^T * scale + bias) V, optionally decoupled into two independently normalised key
// segments blended as seg1 + scale2 * seg2 -- the arithmetic of IPAttnProcessor2_0 (text branch over the frozen
// to_k/to_v, audio branch over to_k_ip/to_v_ip, reference attention_processor.py:429-454) in ONE kernel that
// shares Q, keeps both softmaxes in registers and writes the blended heads once.
//
// Mapping (wave64, MFMA 32x32x16, fp32 accumulate):
//   * one wave owns 32 queries of one (batch, head); 4 waves per workgroup -> 128 queries per workgroup
//   * K / V^T tiles of 64 keys are staged ONCE per workgroup in LDS (coalesced 16-byte global loads issued one tile
//     ahead into registers, written after the MFMAs of the current tile: global latency hides under compute) and
//     shared by the 4 waves; row strides are padded to an odd number of 16-byte slots, so the ds_read_b128 fragment reads (K rows;
//     V^T rows, whose keys are stored permuted inside every 16-key step: Lay::VROW) are bank-conflict free
//   * scores are computed TRANSPOSED, S^T = K . Q^T, so that after the MFMA each lane holds 32 keys of ONE query
//     (column = lane&31): the softmax max/sum are in-lane reductions plus one cross-half exchange per 64 keys
//   * P^T stays in registers: the C-layout of S^T (keys (r&3)+8*(r>>2)+4*half) is re-used directly as the B operand
//     of O^T = V^T . P^T by reading V^T with the SAME key permutation (two 8-byte pieces per lane) -- no LDS round
//     trip and no cross-lane shuffle between the two MFMAs
//   * online softmax in the exp2 domain with the scale folded into one FMA per score; the O rescale is skipped
//     (wave-uniform branch) whenever no lane's running max moved
// V must be supplied transposed per head, [Bk][H][D][Lpad] with zero padding (apad_gemm / apad_rowpanel_gemm
// APAD_OUT_VT).
#include <stdlib.h>
#include "common.h"
#include "f32_ops.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct AttnP {
    const uint8_t* q;
    const uint8_t* k;
    const uint8_t* vt;
    const uint8_t* k2;
    const uint8_t* vt2;
    uint8_t* out;
    const float* key_bias;
    float* lse;
    int64_t q_sb, q_sn, k_sb, k_sl, vt_sb, k2_sb, k2_sl, vt2_sb, o_sb, o_sn;
    int32_t B, N, H, L, Lpad, L2, Lpad2, kvdiv, kvdiv2;
    float scale_log2, scale2;
    int32_t prescaled;  // q already carries softmax_scale * log2(e) (scale_log2 is then exactly 1)
};

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr int KT = 64;  // keys per LDS tile
#ifndef APAD_ABL
#define APAD_ABL 0  // ablation probes of the key loop (tools/ab_build.sh builds only): 1 no exp, 2 no score MFMAs, 4 no P.V MFMAs, 8 no staging
                    // (two-tile kernel also: 16 no barrier (staging kept), 32 no LDS fragment reads, 64 no sums / range check)
#endif


template <int D> struct Lay {
    static constexpr int KROW = (D + 8) * 2;           // K tile row stride (bytes): D/8 + 1 sixteen-byte slots (odd)
    static constexpr int K_BYTES = KT * KROW;
    static constexpr int DT_TILES = (D + 31) / 32;
    static constexpr int VROWS = DT_TILES * 32;        // rows >= D are zeroed once; they feed discarded output rows
    static constexpr int VROW = (KT + 8) * 2;          // V^T tile row stride (bytes): 9 sixteen-byte slots (odd).  Inside every 16-key step the keys are stored
                                                       // as 0-3, 8-11, 4-7, 12-15: the 8 keys a half-wave multiplies with the C-layout probabilities
                                                       // (4 half + {0..3}, 8 + 4 half + {0..3}) are ONE 16-byte read (a ds_read holds a wave's issue ~18
                                                       // cycles whatever its width: two 8-byte reads per fragment cost twice that)
    static constexpr int V_BYTES = VROWS * VROW;
    static constexpr int BUF = K_BYTES + V_BYTES;
    static constexpr int KCH = KT * (D / 8);           // 16-byte chunks of a K tile
    static constexpr int VCH = D * (KT / 8);           // 16-byte chunks of a V^T tile
    static constexpr int NK = (KCH + 255) / 256;       // staging registers per thread
    static constexpr int NV = (VCH + 255) / 256;
};

// global -> registers for key tile starting at key0 (rows / columns outside the segment read as zero)
template <int D>
__device__ __forceinline__ void tile_load(u32x4 (&rk)[Lay<D>::NK], u32x4 (&rv)[Lay<D>::NV], const uint8_t* kbase, int64_t k_sl,
                                          const uint8_t* vbase, int L, int Lpad, int key0, int tid) {
    using Y = Lay<D>;
#pragma unroll
    for (int i = 0; i < Y::NK; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / (D / 8), ch = idx - row * (D / 8);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (idx < Y::KCH && key0 + row < L) v = *reinterpret_cast<const u32x4*>(kbase + ((int64_t)(key0 + row) * k_sl + ch * 8) * 2);
        rk[i] = v;
    }
#pragma unroll
    for (int i = 0; i < Y::NV; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 3, ch = idx & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (idx < Y::VCH && key0 + ch * 8 < Lpad) v = *reinterpret_cast<const u32x4*>(vbase + ((int64_t)row * Lpad + key0 + ch * 8) * 2);
        rv[i] = v;
    }
}

// same for a tile that lies entirely inside the segment: per-thread byte offsets are precomputed once per segment and the
// tile base is wave-uniform, so a tile costs no address arithmetic and no predicates on the vector ALU (the generic loader
// spent ~25 VALU instructions per tile on them, in a loop that is VALU-bound)
template <int D>
__device__ __forceinline__ void tile_load_full(u32x4 (&rk)[Lay<D>::NK], u32x4 (&rv)[Lay<D>::NV], const uint8_t* ktile,
                                               const uint8_t* vtile, const uint32_t (&koff)[Lay<D>::NK],
                                               const uint32_t (&voff)[Lay<D>::NV], int tid) {
    using Y = Lay<D>;
#pragma unroll
    for (int i = 0; i < Y::NK; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (tid + 256 * i < Y::KCH) v = *reinterpret_cast<const u32x4*>(ktile + koff[i]);
        rk[i] = v;
    }
#pragma unroll
    for (int i = 0; i < Y::NV; ++i) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (tid + 256 * i < Y::VCH) v = *reinterpret_cast<const u32x4*>(vtile + voff[i]);
        rv[i] = v;
    }
}

template <int D>
__device__ __forceinline__ void tile_store(const u32x4 (&rk)[Lay<D>::NK], const u32x4 (&rv)[Lay<D>::NV], uint8_t* buf, int tid) {
    using Y = Lay<D>;
#pragma unroll
    for (int i = 0; i < Y::NK; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / (D / 8), ch = idx - row * (D / 8);
        if (idx < Y::KCH) *reinterpret_cast<u32x4*>(buf + row * Y::KROW + ch * 16) = rk[i];
    }
#pragma unroll
    for (int i = 0; i < Y::NV; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 3, ch = idx & 7;
        if (idx < Y::VCH) {  // keys 8 ch .. 8 ch + 7 of the row: two 8-byte stores into the permuted 16-key step (Lay::VROW)
            u32x2 lo = {rv[i][0], rv[i][1]}, hi = {rv[i][2], rv[i][3]};
            uint8_t* dst = buf + Y::K_BYTES + row * Y::VROW + (ch >> 1) * 32 + (ch & 1) * 8;
            *reinterpret_cast<u32x2*>(dst) = lo;
            *reinterpret_cast<u32x2*>(dst + 16) = hi;
        }
    }
}

// Scores, softmax and P.V for ONE staged 64-key tile.  MASK = false is the steady-state path (all 64 keys valid, no
// bias): max on the raw scores, one packed FMA per two scores folds scale and running max into the exp2 argument.
// MASK = true handles the additive key bias and the ragged last tile.  Softmax denominators: every lane owns one query
// (half of its keys), so they are a per-lane packed-fp32 sum (osum, two partial sums; the halves meet once per segment).
// An earlier version put them on the matrix pipe (one MFMA per 16-key step against a "ones" fragment); MFMA and vector
// work do not overlap inside a wave (tools/probes/overlap.hip), so 16 v_pk_add_f32 (64 cycles) beat 4 MFMAs (128):
// self-attention N=1000 147.5 -> 142.8 us.
template <int DT, int D, bool MASK>
__device__ __forceinline__ void tile_compute(const uint8_t* buf, int key0, int L, const float* bias, float c,
                                             const typename ET<DT>::v8* qf, f32x16* o, f32x2& osum, float& m, int l31,
                                             int half) {
    using E = ET<DT>;
    using Y = Lay<D>;
    constexpr int KC = D / 16;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // ---- S^T (two 32-key sub-tiles) = K . Q^T ----
    // short segments (8 text tokens, 32 audio tokens, 16 T5 tokens: the decoupled cross-attention of the adapter) fill
    // at most the first 32-key sub-tile of their only tile: the second sub-tile's MFMAs, exponentials and P.V steps are
    // skipped (wave-uniform), which is half the work of such a launch
    const bool one_sub = MASK && (L - key0) <= 32;
    f32x16 s[2];
    s[1] = zero16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u == 1 && one_sub) break;
        const uint8_t* kp = buf + (u * 32 + l31) * Y::KROW + half * 16;
        if (APAD_ABL & 2) {
            const uint4 kq = *reinterpret_cast<const uint4*>(kp);
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][r] = __uint_as_float(kq.x + r) * 1e-30f;
        } else {
        s[u] = E::mfma32(as_v8<DT>(*reinterpret_cast<const uint4*>(kp)), qf[0], zero16);
#pragma unroll
        for (int cc = 1; cc < KC; ++cc) {
            typename E::v8 kf = as_v8<DT>(*reinterpret_cast<const uint4*>(kp + cc * 32));
            s[u] = E::mfma32(kf, qf[cc], s[u]);
        }
        }
    }
    float tmax = NEG_BIG;
    if (MASK) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && one_sub) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key0 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = s[u][r] * c;
                if (bias) v += bias[key < L ? key : L - 1] * LOG2E;
                v = key < L ? v : NEG_BIG;
                s[u][r] = v;
                tmax = fmaxf(tmax, v);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[u][r]);
        tmax *= c;  // c > 0
    }
    tmax = half_max(tmax);
    const float mnew = fmaxf(m, tmax);
    if (__any(mnew > m)) {  // wave-uniform: the O rescale is skipped whenever no lane's running max moved
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        osum *= alpha;
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        m = mnew;
    }
    if (MASK) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && one_sub) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[u][r] = __builtin_amdgcn_exp2f(s[u][r] - m);
                osum[0] += s[u][r];
            }
        }
    } else {
        // four independent partial sums: one accumulator chained 16 dependent v_pk_add_f32, each followed by a hazard s_nop
        const f32x2 c2 = {c, c}, nm2 = {-m, -m};
        f32x2 part[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v = {s[u][r], s[u][r + 1]};
                v = __builtin_elementwise_fma(v, c2, nm2);
                s[u][r] = (APAD_ABL & 1) ? v[0] * 0.5f : __builtin_amdgcn_exp2f(v[0]);
                s[u][r + 1] = (APAD_ABL & 1) ? v[1] * 0.5f : __builtin_amdgcn_exp2f(v[1]);
                part[(r >> 1) & 3] += (f32x2){s[u][r], s[u][r + 1]};
            }
        osum += (part[0] + part[1]) + (part[2] + part[3]);
    }
    // ---- O^T += V^T . P^T : four K=16 steps; B operand = P^T in its C-layout key order ----
    const uint8_t* vtile = buf + Y::K_BYTES;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st == 2 && one_sub) break;
        typename E::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[st >> 1][(st & 1) * 8 + j];
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt) {
            typename E::v8 vf = as_v8<DT>(*reinterpret_cast<const uint4*>(vtile + (dt * 32 + l31) * Y::VROW + (st * 16 + 8 * half) * 2));
            if (APAD_ABL & 4) o[dt][st] += (float)vf[0] * (float)pf[0];
            else o[dt] = E::mfma32(vf, pf, o[dt]);
        }
    }
}

// One softmax segment over L keys: accumulates un-normalised O^T into o[] and the denominators into osum; m is the
// running max in the scaled log2 domain.  All 256 threads of the workgroup must call it (LDS staging).
template <int DT, int D>
__device__ __forceinline__ void segment(uint8_t* smem, const uint8_t* kbase, int64_t k_sl, const uint8_t* vbase, int L, int Lpad,
                                        const float* bias, float c, const typename ET<DT>::v8* qf, f32x16* o, f32x2& osum,
                                        float& m, int tid) {
    using Y = Lay<D>;
    const int lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int ntiles = (L + KT - 1) / KT;
    const int nfull = bias ? 0 : L / KT;  // tiles that need neither bias nor tail masking
    u32x4 rk[Y::NK], rv[Y::NV];
    uint32_t koff[Y::NK], voff[Y::NV];  // per-thread byte offsets inside a tile (rows x k_sl stays far below 4 GB)
#pragma unroll
    for (int i = 0; i < Y::NK; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / (D / 8), ch = idx - row * (D / 8);
        koff[i] = (uint32_t)(((int64_t)row * k_sl + ch * 8) * 2);
    }
#pragma unroll
    for (int i = 0; i < Y::NV; ++i) {
        const int idx = tid + 256 * i;
        voff[i] = (uint32_t)((((int64_t)(idx >> 3)) * Lpad + (idx & 7) * 8) * 2);
    }
    tile_load<D>(rk, rv, kbase, k_sl, vbase, L, Lpad, 0, tid);
    __syncthreads();  // previous users of the LDS buffers (other segment / zero fill) are done
    tile_store<D>(rk, rv, smem, tid);
    __syncthreads();
    // two loops instead of one loop with a per-tile branch: the accumulators then live in fixed registers inside
    // each loop (a merged loop made the compiler copy o/osum around every tile)
    int t = 0;
    for (; t < nfull; ++t) {
        const uint8_t* buf = smem + (t & 1) * Y::BUF;
        if (t + 1 < nfull)  // the next tile is a full one too
            tile_load_full<D>(rk, rv, kbase + (int64_t)(t + 1) * KT * k_sl * 2, vbase + (int64_t)(t + 1) * KT * 2, koff, voff, tid);
        else if (t + 1 < ntiles) tile_load<D>(rk, rv, kbase, k_sl, vbase, L, Lpad, (t + 1) * KT, tid);
        tile_compute<DT, D, false>((APAD_ABL & 8) ? smem : buf, t * KT, L, bias, c, qf, o, osum, m, l31, half);
        if (!(APAD_ABL & 8)) {
        if (t + 1 < ntiles) tile_store<D>(rk, rv, smem + ((t + 1) & 1) * Y::BUF, tid);
        __syncthreads();
        }
    }
    for (; t < ntiles; ++t) {
        const uint8_t* buf = smem + (t & 1) * Y::BUF;
        if (t + 1 < ntiles) tile_load<D>(rk, rv, kbase, k_sl, vbase, L, Lpad, (t + 1) * KT, tid);
        tile_compute<DT, D, true>(buf, t * KT, L, bias, c, qf, o, osum, m, l31, half);
        if (t + 1 < ntiles) tile_store<D>(rk, rv, smem + ((t + 1) & 1) * Y::BUF, tid);
        __syncthreads();
    }
}

template <int DT, int D, bool DUAL>
__global__ __launch_bounds__(256) void attn_kernel(AttnP p) {
    using E = ET<DT>;
    using Y = Lay<D>;
    constexpr int KC = D / 16;
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * Y::BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // 1-D grid ordered for the 8 XCD-private L2s: workgroup id w runs on XCD w % 8 (observed, speed only).  Consecutive
    // ids on ONE XCD walk the query tiles of ONE (batch, head), so its K/V are fetched into that L2 once and reused
    // while hot.
    const int nbh = p.B * p.H;
    const int nqt = (p.N + 127) >> 7;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int bh = (seq / nqt) * 8 + xcd, qt = seq % nqt;
    if (bh >= nbh) return;
    const int h = bh % p.H, b = bh / p.H;
    const int q0 = qt * 128 + wave * 32;
    int qi = q0 + l31;
    const bool qvalid = qi < p.N;
    qi = qvalid ? qi : p.N - 1;

    // zero the V^T tile rows >= D once (they are read as A-operand rows whose outputs are discarded; keep them finite)
    if (Y::VROWS > D) {
        for (int i = tid; i < 2 * Y::BUF / 4; i += 256) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    }

    // Q^T B-operand fragments: lane holds Q[qi][c*16 + half*8 .. +8)
    typename E::v8 qf[KC];
    const uint8_t* qp = p.q + ((int64_t)b * p.q_sb + (int64_t)qi * p.q_sn + h * D + half * 8) * 2;
#pragma unroll
    for (int cc = 0; cc < KC; ++cc) qf[cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(qp + cc * 32));

    f32x16 o[Y::DT_TILES];
#pragma unroll
    for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x2 osum = {0.f, 0.f};
    float m = NEG_BIG;

    {
        const int bk = b / p.kvdiv;
        const uint8_t* kbase = p.k + ((int64_t)bk * p.k_sb + h * D) * 2;
        const uint8_t* vbase = p.vt + ((int64_t)bk * p.vt_sb + (int64_t)h * D * p.Lpad) * 2;
        const float* bias = p.key_bias ? p.key_bias + (int64_t)b * p.L : nullptr;
        segment<DT, D>(smem, kbase, p.k_sl, vbase, p.L, p.Lpad, bias, p.scale_log2, qf, o, osum, m, tid);
    }
    // denominator of query l31: the two partial sums of both half-wave lanes that own it
    const float den = half_sum(osum[0] + osum[1]);
    float inv = 1.0f / den;
    // log2 sum exp2 of the scaled (+biased) scores: what the backward re-uses.  The pad entries [N, round_up(N, 32)) are written too
    // (0): the backward multiplies exp2(s - lse) of padded queries by zero-padded operands, so they must be finite, and the caller
    // need not pre-fill the buffer
    if (p.lse != nullptr && half == 0 && q0 + l31 < ((p.N + 31) & ~31))
        p.lse[((int64_t)b * p.H + h) * ((p.N + 31) & ~31) + q0 + l31] = qvalid ? m + __builtin_log2f(den) : 0.f;
#pragma unroll
    for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= inv;

    if (DUAL) {
        if (p.L2 > 0) {
            f32x16 o2[Y::DT_TILES];
#pragma unroll
            for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o2[dt][r] = 0.f;
            f32x2 osum2 = {0.f, 0.f};
            float m2 = NEG_BIG;
            const int bk = b / p.kvdiv2;
            const uint8_t* kbase = p.k2 + ((int64_t)bk * p.k2_sb + h * D) * 2;
            const uint8_t* vbase = p.vt2 + ((int64_t)bk * p.vt2_sb + (int64_t)h * D * p.Lpad2) * 2;
            segment<DT, D>(smem, kbase, p.k2_sl, vbase, p.L2, p.Lpad2, nullptr, p.scale_log2, qf, o2, osum2, m2, tid);
            const float inv2 = 1.0f / half_sum(osum2[0] + osum2[1]);
            // the un-fused reference rounds each branch, and scale * audio, to the storage type before the add
#pragma unroll
            for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = (float)(typename E::elem)o[dt][r];
                    const float a = (float)(typename E::elem)(o2[dt][r] * inv2);
                    o[dt][r] = t + (float)(typename E::elem)(p.scale2 * a);
                }
        }
    }

    // ---- store: transpose through LDS (the K/V buffers are free after the last tile's barrier) so that each store
    //      instruction writes whole D*2-byte row segments with 16 bytes per lane instead of 32 rows x 16 bytes ----
    constexpr int OROW = D * 2 + 8;  // odd number of 8-byte slots per row
    uint8_t* scr = smem + wave * (32 * OROW);
#pragma unroll
    for (int dt = 0; dt < Y::DT_TILES; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dcol = dt * 32 + 8 * g + 4 * half;
            if (dcol < D) {
                typename E::v4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)o[dt][g * 4 + j];
                *reinterpret_cast<uint2*>(scr + l31 * OROW + dcol * 2) = __builtin_bit_cast(uint2, pk);
            }
        }
    }
    constexpr int CPR = D / 8;  // 16-byte chunks per row
    uint8_t* ob = p.out + ((int64_t)b * p.o_sb + h * D) * 2;
    for (int idx = lane; idx < 32 * CPR; idx += 64) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const int q = q0 + row;
        if (q < p.N) {
            const uint2 lo = *reinterpret_cast<const uint2*>(scr + row * OROW + ch * 16);
            const uint2 hi = *reinterpret_cast<const uint2*>(scr + row * OROW + ch * 16 + 8);
            *reinterpret_cast<uint4*>(ob + ((int64_t)q * p.o_sn + ch * 8) * 2) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Long self-attention variant: a wave owns TWO 32-query tiles (64 queries; 256 per workgroup).  Inside one wave MFMAs
// only hide under VALU work of the SAME wave that does not depend on them (tools/probes/pair.hip), and with one query
// tile the key loop is one dependency chain (scores -> softmax -> P.V).  Two tiles give the instruction stream independent
// work: tile B's score MFMAs run under tile A's softmax, tile A's P.V MFMAs under tile B's softmax; each K fragment read
// from LDS feeds two MFMAs and the staging / barrier cost per query halves.  No key bias, one segment (the UNet's attn1).
//
// DIRECT (q pre-scaled by softmax_scale * log2 e, full tiles): the softmax costs exp + sum + pack per score and nothing else.
//   * the running max enters through the score MFMA's C operand: mi[qt] holds -m in all 16 accumulator registers, so the MFMA
//     returns s - m and the exponential is taken of the accumulator as it is (no scale / subtract instruction per score);
//   * no per-tile max either: probabilities are allowed to exceed 1.  bf16 / fp32 carry 8 exponent bits, so a score that runs
//     up to log2(BIG) above the running max is harmless; the tile's probability SUM (needed anyway) is the range check, and only
//     when some lane's sum leaves [0, BIG) -- the first tile, or a genuine jump of the maximum -- the wave takes the classic path
//     (raw scores, new maximum, rescale of O and the denominators) for that tile.  Exact arithmetic otherwise: the result is a
//     softmax with a different, equally valid, reference point.
template <int DT, int D, bool MASK, bool DIRECT = false>
__device__ __forceinline__ void tile_compute2(const uint8_t* buf, int key0, int L, float c, const typename ET<DT>::v8 (&qf)[2][D / 16],
                                              f32x16 (&o)[2][Lay<D>::DT_TILES], f32x2 (&osum)[2], float (&m)[2], f32x16 (&mi)[2], int l31, int half) {
    using E = ET<DT>;
    using Y = Lay<D>;
    constexpr int KC = D / 16;
    // largest tile sum the fast path accepts: P is rounded to the storage type for the P.V product (f16 overflows at 65504)
    constexpr float BIG = (DT == APAD_F16) ? 4096.f : 1073741824.f;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[2][2];  // [query tile][32-key sub-tile]
    // every fragment of the tile is requested from LDS up front, in one batch: K (A operand of both tiles' score MFMAs) and V^T
    // (A operand of both tiles' P.V MFMAs).  Left inline, each read was waited for right before its MFMA (4 exposed LDS round
    // trips per tile).
    typename E::v8 kf[2][KC], vf[4][Y::DT_TILES];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) {
            if (APAD_ABL & 32) kf[u][cc] = qf[0][cc];
            else
            kf[u][cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(buf + (u * 32 + l31) * Y::KROW + half * 16 + cc * 32));
        }
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt) {
            if (APAD_ABL & 32) { vf[st][dt] = qf[1][0]; continue; }
            vf[st][dt] = as_v8<DT>(*reinterpret_cast<const uint4*>(buf + Y::K_BYTES + (dt * 32 + l31) * Y::VROW + (st * 16 + 8 * half) * 2));
        }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (APAD_ABL & 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qt][u][r] = mi[qt][r] * 1e-30f + (float)kf[u][0][r & 7] * (float)qf[qt][0][r & 7];
                continue;
            }
            s[qt][u] = E::mfma32(kf[u][0], qf[qt][0], (DIRECT && !MASK) ? mi[qt] : zero16);
#pragma unroll
            for (int cc = 1; cc < KC; ++cc) s[qt][u] = E::mfma32(kf[u][cc], qf[qt][cc], s[qt][u]);
        }
    // pin: the V^T fragments are complete HERE (behind the score MFMAs, where the wait is free) -- without it the compiler sinks
    // each read to just before its P.V MFMA and waits for it there
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt) asm volatile("" : "+v"(vf[st][dt]));
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if constexpr (DIRECT && !MASK) {
            f32x2 part[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    s[qt][u][r] = (APAD_ABL & 1) ? s[qt][u][r] * 0.5f : __builtin_amdgcn_exp2f(s[qt][u][r]);
                    s[qt][u][r + 1] = (APAD_ABL & 1) ? s[qt][u][r + 1] * 0.5f : __builtin_amdgcn_exp2f(s[qt][u][r + 1]);
                    if (!(APAD_ABL & 64)) part[(r >> 1) & 3] += (f32x2){s[qt][u][r], s[qt][u][r + 1]};
                }
            f32x2 ts = (part[0] + part[1]) + (part[2] + part[3]);
            if (!(APAD_ABL & 64) && __any(!(ts[0] + ts[1] < BIG))) {  // wave-uniform, rare: first tile / the maximum jumped by more than log2(BIG)
                float tmax = NEG_BIG;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    s[qt][u] = E::mfma32(kf[u][0], qf[qt][0], zero16);
#pragma unroll
                    for (int cc = 1; cc < KC; ++cc) s[qt][u] = E::mfma32(kf[u][cc], qf[qt][cc], s[qt][u]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qt][u][r]);
                }
                const float mnew = fmaxf(m[qt], half_max(tmax));
                const float alpha = __builtin_amdgcn_exp2f(m[qt] - mnew);
                osum[qt] *= alpha;
#pragma unroll
                for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
                m[qt] = mnew;
#pragma unroll
                for (int r = 0; r < 16; ++r) mi[qt][r] = -mnew;
                ts = (f32x2){0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        s[qt][u][r] = __builtin_amdgcn_exp2f(s[qt][u][r] - mnew);
                        s[qt][u][r + 1] = __builtin_amdgcn_exp2f(s[qt][u][r + 1] - mnew);
                        ts += (f32x2){s[qt][u][r], s[qt][u][r + 1]};
                    }
            }
            osum[qt] += ts;
        } else {
        float tmax = NEG_BIG;
        if (MASK) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = key < L ? s[qt][u][r] * c : NEG_BIG;
                    s[qt][u][r] = v;
                    tmax = fmaxf(tmax, v);
                }
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[qt][u][r]);
            tmax *= c;  // c > 0
        }
        tmax = half_max(tmax);
        const float mnew = fmaxf(m[qt], tmax);
        // rescale only when some lane's running max moved (wave-uniform; rare after the first tiles).  MFMA and VALU time add up
        // on this part whatever the interleave (NOTES §4b), so the branch costs nothing and the skipped multiplies are a net gain
        if (__any(mnew > m[qt])) {
            const float alpha = __builtin_amdgcn_exp2f(m[qt] - mnew);
            osum[qt] *= alpha;
#pragma unroll
            for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
            m[qt] = mnew;
        }
        if (MASK) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[qt][u][r] = __builtin_amdgcn_exp2f(s[qt][u][r] - mnew);
                    osum[qt][0] += s[qt][u][r];
                }
        } else {
            // four independent partial sums per sub-tile pair: a single accumulator made every v_pk_add_f32 wait on the previous
            // one (each pair cost a hazard s_nop on top of its issue slot)
            const f32x2 c2 = {c, c}, nm2 = {-mnew, -mnew};
            f32x2 part[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 v = {s[qt][u][r], s[qt][u][r + 1]};
                    v = __builtin_elementwise_fma(v, c2, nm2);
                    s[qt][u][r] = __builtin_amdgcn_exp2f(v[0]);
                    s[qt][u][r + 1] = __builtin_amdgcn_exp2f(v[1]);
                    part[(r >> 1) & 3] += (f32x2){s[qt][u][r], s[qt][u][r + 1]};
                }
            osum[qt] += (part[0] + part[1]) + (part[2] + part[3]);
        }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            typename E::v8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[qt][st >> 1][(st & 1) * 8 + j];
#pragma unroll
            for (int dt = 0; dt < Y::DT_TILES; ++dt) {
                if (APAD_ABL & 4) o[qt][dt][st] += (float)vf[st][dt][0] * (float)pf[0] + (float)pf[7];
                else
                o[qt][dt] = E::mfma32(vf[st][dt], pf, o[qt][dt]);
            }
        }
    }
}

// Measured and removed (round 3; tools/ab_build.sh ablations of the direct loop, isolated, 64 x 8 heads x 1000 x 1000, 128.7 us):
// without the exponentials 120.2, without the P.V MFMAs 117.0, without staging 114.6 (the barrier alone: 128.0), without the LDS
// fragment reads 103.9, without staging AND fragment reads 83.2, without the denominators / range check 100.1, skeleton 26.9.
//   * a three-stage LDS ring with the next tile's K fragments requested behind the score MFMAs, scalar-add denominators and one
//     range check per tile pair: 130.8 us (no gain -- the exposed latencies are not the fragment reads');
//   * the same ring with a BRANCH-FREE key loop (tile 0 classic to fix the reference point, a sticky out-of-window flag, classic
//     recomputation of a flagged workgroup after the loop): 122 us with the recomputation compiled out, but 154 us with it
//     inlined (+23 VGPRs, 6 scalar spills in the hot loop) and worse as a noinline call (scratch + 16 scalar spills).  A 5 % gain
//     that needs a second launch per attention call for the flagged workgroups was not worth it.
// An inline-asm v_add_f32 for the denominators read STALE registers: hipcc pads no hazards around asm, and a v_exp_f32 result
// consumed by the next VALU instruction needs a wait state (caught by tests/test_gpu_kernels.py::test_attention_prescaled_q).
// NW = waves per workgroup: 4 (256 queries) or, for d = 32, 8 (512 queries: the K / V^T tile of a (batch, head) is staged once
// per CU instead of twice; waves 0-3 stage K, waves 4-7 stage V^T, one 16-byte chunk per thread)
template <int DT, int D, int NW, bool DIRECT>
__device__ __forceinline__ void attn2q_body(const AttnP& p, uint8_t* smem) {
    using E = ET<DT>;
    using Y = Lay<D>;
    constexpr int KC = D / 16, QPW = NW * 64;
    static_assert(NW == 4 || (NW == 8 && D == 32), "the 8-wave staging split is written for d = 32");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nbh = p.B * p.H;
    const int nqt = (p.N + QPW - 1) / QPW;
    // XCD-aware order (workgroup id w runs on XCD w % 8: observed, speed only): an XCD walks whole SAMPLES -- all query tiles of
    // head 0, then head 1, ... of sample b = 8 i + xcd.  The query tiles of one (sample, head) share its K / V^T through that XCD's
    // L2, and so do NEIGHBOURING HEADS: a 128-byte line of K [B][L][H*32] holds the 64-byte rows of two heads, so with heads spread
    // over the XCDs (the earlier order) every K line was fetched by two L2s (PMC: 164.8 MB read per launch against 98.3 MB)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int grp = seq / nqt, qtile = seq % nqt;
    const int b = (grp / p.H) * 8 + xcd, h = grp % p.H;
    if (b >= p.B) return;
    (void)nbh;
    const int q0 = qtile * QPW + wave * 64;
    if (Y::VROWS > D) {
        for (int i = tid; i < 2 * Y::BUF / 4; i += NW * 64) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    }
    typename E::v8 qf[2][KC];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int qi = q0 + qt * 32 + l31;
        qi = qi < p.N ? qi : p.N - 1;
        const uint8_t* qp = p.q + ((int64_t)b * p.q_sb + (int64_t)qi * p.q_sn + h * D + half * 8) * 2;
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) qf[qt][cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(qp + cc * 32));
    }
    f32x16 o[2][Y::DT_TILES];
    f32x2 osum[2];
    float m[2];
    f32x16 mi[2];  // DIRECT: -m in every accumulator register of a score tile (the score MFMA's C operand)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        osum[qt] = (f32x2){0.f, 0.f};
        m[qt] = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) mi[qt][r] = -NEG_BIG;  // exp2(s + 1e30) = inf: the first tile takes the classic path
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qt][dt][r] = 0.f;
    }
    const int bk = b / p.kvdiv;
    const uint8_t* kbase = p.k + ((int64_t)bk * p.k_sb + h * D) * 2;
    const uint8_t* vbase = p.vt + ((int64_t)bk * p.vt_sb + (int64_t)h * D * p.Lpad) * 2;
    const int L = p.L, Lpad = p.Lpad;
    const int64_t k_sl = p.k_sl;
    const float c = p.scale_log2;
    {   // key loop: the staging of segment(), two query tiles per wave
        const int ntiles = (L + KT - 1) / KT, nfull = L / KT;
        u32x4 rk[Y::NK], rv[Y::NV];
        uint32_t koff[Y::NK], voff[Y::NV];
        // 8-wave staging: this thread's ONE chunk of the tile -- K row (t2 / 4), chunk (t2 % 4) for waves 0-3; V^T row (t2 / 8),
        // chunk (t2 % 8) for waves 4-7
        const bool isK = tid < 256;
        const int t2 = tid & 255;
        const int srow = isK ? (t2 >> 2) : (t2 >> 3), sch = isK ? (t2 & 3) : (t2 & 7);
        const uint32_t soff8 = isK ? (uint32_t)(((int64_t)srow * k_sl + sch * 8) * 2) : (uint32_t)(((int64_t)srow * Lpad + sch * 8) * 2);
        u32x4 sreg = {0u, 0u, 0u, 0u};
        if constexpr (NW == 4) {
#pragma unroll
            for (int i = 0; i < Y::NK; ++i) {
                const int idx = tid + 256 * i;
                const int row = idx / (D / 8), ch = idx - row * (D / 8);
                koff[i] = (uint32_t)(((int64_t)row * k_sl + ch * 8) * 2);
            }
#pragma unroll
            for (int i = 0; i < Y::NV; ++i) {
                const int idx = tid + 256 * i;
                voff[i] = (uint32_t)((((int64_t)(idx >> 3)) * Lpad + (idx & 7) * 8) * 2);
            }
        }
        auto load_tile = [&](int tile, bool full) {
            const int key0 = tile * KT;
            if constexpr (NW == 4) {
                if (full) tile_load_full<D>(rk, rv, kbase + (int64_t)key0 * k_sl * 2, vbase + (int64_t)key0 * 2, koff, voff, tid);
                else tile_load<D>(rk, rv, kbase, k_sl, vbase, L, Lpad, key0, tid);
            } else {
                const u32x4 z = {0u, 0u, 0u, 0u};
                if (isK) sreg = (full || key0 + srow < L) ? *reinterpret_cast<const u32x4*>(kbase + (int64_t)key0 * k_sl * 2 + soff8) : z;
                else sreg = (full || key0 + sch * 8 < Lpad) ? *reinterpret_cast<const u32x4*>(vbase + (int64_t)key0 * 2 + soff8) : z;
            }
        };
        auto store_tile = [&](uint8_t* buf) {
            if constexpr (NW == 4) {
                tile_store<D>(rk, rv, buf, tid);
            } else {
                if (isK) {
                    *reinterpret_cast<u32x4*>(buf + srow * Y::KROW + sch * 16) = sreg;
                } else {  // keys 8 sch .. 8 sch + 7 of the row: two 8-byte stores into the permuted 16-key step (Lay::VROW)
                    uint8_t* dst = buf + Y::K_BYTES + srow * Y::VROW + (sch >> 1) * 32 + (sch & 1) * 8;
                    *reinterpret_cast<u32x2*>(dst) = (u32x2){sreg[0], sreg[1]};
                    *reinterpret_cast<u32x2*>(dst + 16) = (u32x2){sreg[2], sreg[3]};
                }
            }
        };
        load_tile(0, false);
        __syncthreads();
        store_tile(smem);
        __syncthreads();
        int t = 0;
        for (; t < nfull; ++t) {
            const uint8_t* buf = smem + (t & 1) * Y::BUF;
            if (!(APAD_ABL & 8)) {
            if (t + 1 < nfull) load_tile(t + 1, true);
            else if (t + 1 < ntiles) load_tile(t + 1, false);
            }
            tile_compute2<DT, D, false, DIRECT>(buf, t * KT, L, c, qf, o, osum, m, mi, l31, half);
            if (!(APAD_ABL & 8)) {
            if (t + 1 < ntiles) store_tile(smem + ((t + 1) & 1) * Y::BUF);
            if (!(APAD_ABL & 16)) __syncthreads();
            }
        }
        for (; t < ntiles; ++t) {
            const uint8_t* buf = smem + (t & 1) * Y::BUF;
            if (t + 1 < ntiles) load_tile(t + 1, false);
            tile_compute2<DT, D, true, DIRECT>(buf, t * KT, L, c, qf, o, osum, m, mi, l31, half);
            if (t + 1 < ntiles) store_tile(smem + ((t + 1) & 1) * Y::BUF);
            __syncthreads();
        }
    }
    constexpr int OROW = D * 2 + 8;
    constexpr int CPR = D / 8;
    uint8_t* scr = smem + wave * (32 * OROW);
    uint8_t* ob = p.out + ((int64_t)b * p.o_sb + h * D) * 2;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const float den = half_sum(osum[qt][0] + osum[qt][1]);
        const float inv = 1.0f / den;
        const int qb = q0 + qt * 32;
        if (p.lse != nullptr && half == 0 && qb + l31 < ((p.N + 31) & ~31))  // (pad entries: 0, see attn_kernel)
            p.lse[((int64_t)b * p.H + h) * ((p.N + 31) & ~31) + qb + l31] = qb + l31 < p.N ? m[qt] + __builtin_log2f(den) : 0.f;
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dcol = dt * 32 + 8 * g + 4 * half;
                if (dcol < D) {
                    typename E::v4 pk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)(o[qt][dt][g * 4 + j] * inv);
                    *reinterpret_cast<uint2*>(scr + l31 * OROW + dcol * 2) = __builtin_bit_cast(uint2, pk);
                }
            }
        }
        for (int idx = lane; idx < 32 * CPR; idx += 64) {
            const int row = idx / CPR, ch = idx - row * CPR;
            const int q = qb + row;
            if (q < p.N) {
                const uint2 lo = *reinterpret_cast<const uint2*>(scr + row * OROW + ch * 16);
                const uint2 hi = *reinterpret_cast<const uint2*>(scr + row * OROW + ch * 16 + 8);
                *reinterpret_cast<uint4*>(ob + ((int64_t)q * p.o_sn + ch * 8) * 2) = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
    }
}

template <int DT, int D, int NW, bool DIRECT = false>
__global__ __launch_bounds__(NW * 64) void attn2q_kernel(AttnP p) {
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * Lay<D>::BUF];
    attn2q_body<DT, D, NW, DIRECT>(p, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// apad_self_attention_fused: LayerNorm + to_q | to_k | to_v + softmax attention of a self-attention sub-layer in ONE launch for the two large
// levels (C = 256 / d = 32 / <= 1024 tokens, C = 384 / d = 48 / <= 256 tokens): the row-panel projection launch (q, k, v^T written to HBM and read
// back: 196 MB at the 1000-token level) and the attention launch become one workgroup = (sample, head) that
//   1. projects the head's K, V^T for ALL tokens of the sample into LDS (in tile_compute2's 64-key tile layout: 16 tiles = 148 KB at 1000 tokens)
//      and its Q for the tokens each wave will attend from -- 32-token panels, wave w owns panels w, w + NW, ...; the head's packed weight
//      fragments (48 KB) stream from L2 / L1 k-step by k-step into registers (NSET k-steps ahead), the token fragments are read RAW straight from
//      global memory in the B-operand layout and the LayerNorm is applied by algebra on the accumulators (statistics summed while the fragments
//      pass: y = rstd (W' x - mean colsum(W')) + W beta, W' = W gamma; apad_gemm's folded LayerNorm) -- nothing of x is staged or kept;
//   2. one barrier, then every wave runs the two-query-tile key loop (tile_compute2, direct form: q carries log2(e) / sqrt(d)) over the RESIDENT
//      K / V^T tiles: no staging, no barrier in the loop; its q fragments never left the registers (the score MFMA contracts over d in the
//      order the projection's C layout produced it: K is stored with the same permutation of d inside every 16-wide k-step).
// to_out + residual stay a separate launch (all heads).  Replaces norm1 + to_q / to_k / to_v + scaled_dot_product_attention of
// attention_processor.py:256-276 behind BasicTransformerBlock.norm1 / norm2 (double self-attention).
struct SfP {
    const uint8_t* x;
    const uint8_t* w;     // packed [H][T3 = ceil(3 d / 32) row tiles over the head's q | k | v rows][KC k-steps][64 lanes][8] (gamma, softmax scale folded in)
    const float* csbb;    // [H][2][T3 * 32] fp32: row sums of the packed weights, then W . beta
    uint8_t* out;         // O [B][N][C]
    int32_t B, N, H;
    float eps;
};
typedef const __attribute__((address_space(1))) uint8_t* sf_gptr;
typedef const __attribute__((address_space(1))) u32x4* sf_gptr16;
__device__ __forceinline__ sf_gptr sf_sgpr_ptr(const uint8_t* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (sf_gptr)(((uint64_t)hi << 32) | lo);
}
#ifndef SF_NSET
#define SF_NSET 0  // (0: per-geometry default)
#endif
#ifndef SF_NSET48
#define SF_NSET48 1  // weight / token fragment register sets of the projection loop at d = 48 (1: one rolling set, re-requested behind the k-step's MFMAs -- what fits
                     // beside 160 accumulator registers when two workgroups share a CU; 2 spills 124 registers there)
#endif
#ifndef SF_STG_NSW
#define SF_STG_NSW 1  // weight-fragment register sets of the staged projection loop (1: one rolling set; 137 us at d = 32 against 145 with 2 -- registers)
#endif
#ifndef SF_STAGE
#define SF_STAGE(D_) true  // token fragments of the projection staged through LDS (sf_project_stg) per head size (false: the gathered loads, for A/B builds)
#endif
#ifndef SF_STAT
#define SF_STAT 1  // row statistics of the projection loop: 0 = shifted sums on the vector ALU (8 x (convert, subtract, add, fma) per fragment), 1 = un-shifted sums on
                   // the dot-product instruction (8 instructions per fragment)
#endif
#ifndef SF_DIRECT
#define SF_DIRECT(D_) ((D_) == 32)  // the key loop's direct form (running max through the score MFMA's C operand: 32 more registers) per head size: at d = 48 /
                                    // 252 tokens (four key tiles) the classic two-tile form keeps the kernel inside 256 registers
#endif
#ifndef SF_OCC
#define SF_OCC 2  // waves per SIMD the fused self-attention kernel is compiled for (1: the compiler's own choice, for A/B builds)
#endif
#ifndef SF_NPP48
#define SF_NPP48 2  // panels projected at once at d = 48 (2: 160 accumulator registers; 47.7 -> 42.4 us at 64 x 252 tokens: every weight fragment feeds two panels)
#endif
#ifndef SF_VPM
#define SF_VPM 9  // vector instructions scheduled behind each MFMA of the projection loop
#endif
#ifndef SF_ABL
#define SF_ABL 0  // timing ablations (tools/ab_build.sh; results are wrong): 1 = no key loop, 2 = no projection, 4 = projection without the statistics, 8 = x fragments loaded once, 16 = weight fragments loaded once, 64 = x loaded as whole rows (same bytes, 8 instead of 32 rows per instruction)
#endif

// probe build (tools/ab_build.sh <tag> attention.hip -DSF_TRACE=<wave>; tools/sf_trace.py): s_memtime at the phase boundaries of one wave of every
// workgroup.  Never part of the product library.
#ifdef SF_TRACE
__device__ unsigned long long sf_trace_buf[1024][16];
#define SF_STAMP(i_)                                                                                      \
    if (lane == 0 && wave == (SF_TRACE) && blockIdx.x < 1024) {                                           \
        sf_trace_buf[blockIdx.x][i_] = __builtin_amdgcn_s_memtime();                                      \
        if ((i_) == 0) sf_trace_buf[blockIdx.x][14] = wall_clock64();                                     \
        if ((i_) == 9) sf_trace_buf[blockIdx.x][15] = wall_clock64();                                     \
    }
#else
#define SF_STAMP(i_)
#endif

// sum / sum of squares of a pair of 16-bit values on the dot-product instruction (v_dot2c_f32_bf16 / _f16): no conversion, two values per instruction
typedef __bf16 sf_bf2 __attribute__((ext_vector_type(2)));
typedef _Float16 sf_h2 __attribute__((ext_vector_type(2)));
template <int DT> __device__ __forceinline__ float sf_dot2(uint32_t a, uint32_t b, float c) {
    if constexpr (DT == APAD_BF16) return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(sf_bf2, a), __builtin_bit_cast(sf_bf2, b), c, false);
    else return __builtin_amdgcn_fdot2(__builtin_bit_cast(sf_h2, a), __builtin_bit_cast(sf_h2, b), c, false);
}
template <int DT> __device__ __forceinline__ uint32_t sf_ones2() { return DT == APAD_BF16 ? 0x3f803f80u : 0x3c003c00u; }

// acc[n][j] += W_tile_j . x_panel_n^T over the KC k-steps (raw x), with the row statistics of the panels summed on the way (shifted by the row's
// first element: both halves of a row use the same shift)
template <int DT, int NT3, int NPP, int KC, int NSET>
__device__ __forceinline__ void sf_project(sf_gptr wb, uint32_t loff, const uint8_t* const (&xrow)[NPP], f32x16 (&acc)[NPP][NT3], float (&ssum)[NPP],
                                           float (&sq)[NPP], float (&shift)[NPP]) {
    using E = ET<DT>;
    static_assert(KC % NSET == 0, "the register sets rotate over the k-steps");
    typename E::v8 wf[NSET][NT3], xf[NSET][NPP];
#define SF_LD(i_, kk_)                                                                                                                \
    _Pragma("unroll") for (int j = 0; j < NT3; ++j) wf[i_][j] = __builtin_bit_cast(typename E::v8, *(sf_gptr16)(wb + (j * KC + ((SF_ABL & 16) ? 0 : (kk_))) * 1024 + loff)); \
    _Pragma("unroll") for (int n = 0; n < NPP; ++n) xf[i_][n] = as_v8<DT>(*reinterpret_cast<const uint4*>(xrow[n] + ((SF_ABL & 8) ? 0 : (kk_)) * ((SF_ABL & 64) ? 1024 : 32)));
#define SF_MM(i_)                                                                                   \
    _Pragma("unroll") for (int n = 0; n < NPP; ++n) {                                               \
        _Pragma("unroll") for (int j = 0; j < NT3; ++j) acc[n][j] = E::mfma32(wf[i_][j], xf[i_][n], acc[n][j]); \
        if (!(SF_ABL & 4) && SF_STAT == 1) _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_) {       \
            const uint32_t pr_ = __builtin_bit_cast(u32x4, xf[i_][n])[w_];                          \
            ssum[n] = sf_dot2<DT>(pr_, sf_ones2<DT>(), ssum[n]);                                    \
            sq[n] = sf_dot2<DT>(pr_, pr_, sq[n]);                                                   \
        }                                                                                           \
        if (!(SF_ABL & 4) && SF_STAT == 0) _Pragma("unroll") for (int e = 0; e < 8; ++e) {          \
            const float d_ = (float)xf[i_][n][e] - shift[n];                                        \
            ssum[n] += d_;                                                                          \
            sq[n] = __builtin_fmaf(d_, d_, sq[n]);                                                  \
        }                                                                                           \
    }
#pragma unroll
    for (int i = 0; i < NSET; ++i) { SF_LD(i, i); }
#pragma unroll
    for (int n = 0; n < NPP; ++n) {
        shift[n] = SF_STAT == 1 ? 0.f : half_lo((float)xf[0][n][0]);
        ssum[n] = 0.f;
        sq[n] = 0.f;
    }
#pragma unroll 1
    for (int kk = 0; kk < KC - NSET; kk += NSET) {
#pragma unroll
        for (int i = 0; i < NSET; ++i) {
            SF_MM(i);
            SF_LD(i, kk + i + NSET);
        }
        // one MFMA, then its share of the statistics arithmetic, in program order: the matrix pipe runs beside vector work only when both
        // sit interleaved in ONE wave's stream (tools/probes/overlap.hip: 474 vs 635 cycles clustered); then the fragments NSET k-steps ahead
#pragma unroll
        for (int i = 0; i < NSET; ++i) {
#pragma unroll
            for (int m_ = 0; m_ < NT3 * NPP; ++m_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (!(SF_ABL & 4)) __builtin_amdgcn_sched_group_barrier(0x002, SF_VPM, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, NT3 + NPP, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NSET; ++i) { SF_MM(i); }
#undef SF_LD
#undef SF_MM
}

// The same contraction with the token fragments STAGED THROUGH LDS (round 6).  A B-operand fragment straight from global memory is 32 rows x 32 bytes: 32 cache
// lines per load instruction, and the vector memory pipe takes about a cycle per line -- with the statistics on the dot-product instruction (SF_STAT) that gather is
// what bounds the projection phase (probe build with whole-row loads, SF_ABL = 64: 144 -> 131 us at d = 32).  Here the wave fetches its 64 rows of a four-k-step
// chunk as whole 128-byte pieces (8 rows per instruction = 8 lines), parks them in a private 8 KB LDS window (128-byte row records, 16-byte slot s of row r at
// s ^ ((r >> 1) & 7): conflict-free for the b128 writes and the fragment reads) and reads the fragments back one k-step ahead; the chunk after next is in flight in
// registers meanwhile.  Same MFMA order, same operands: bit-equal to sf_project.  The window aliases key tiles (sf_go picks the instantiation only where no finished
// round has written them; the kernel fences it from the round's own epilogue).
template <int DT, int NT3, int NPP, int KC, int KCH>
__device__ __forceinline__ void sf_project_stg(sf_gptr wb, uint32_t loff, const uint8_t* xb, const int (&pan)[NPP], int N, uint8_t* stg, int lane,
                                               f32x16 (&acc)[NPP][NT3], float (&ssum)[NPP], float (&sq)[NPP], float (&shift)[NPP]) {
    using E = ET<DT>;
    // KCH k-steps per chunk: 4 = 128-byte row records (8 rows = 8 lines per load instruction, 32 staging registers, slot swizzle (row >> 1) & 7);
    //                        2 = 64-byte records (16 rows per instruction, 16 registers, (row >> 2) & 3) -- what fits beside 160 accumulator registers at d = 48
    constexpr int C = KC * 16, BR = KCH * 32, LPR = BR / 16, RPI = 64 / LPR, NI = NPP * 32 / RPI, NCH = KC / KCH, NSW = SF_STG_NSW;
    constexpr int SWS = KCH == 4 ? 1 : 2, SWM = LPR - 1;
    static_assert(KC % KCH == 0 && (KCH == 4 || KCH == 2) && KCH % NSW == 0, "64- / 128-byte row records; the weight sets rotate inside a chunk");
    const int half = lane >> 5, l31 = lane & 31;
    // lane -> (row lane / 8 of an 8-row group, 16-byte piece lane % 8); the per-instruction offsets are re-derived where they are used (registers)
    const int rsub = lane / LPR, piece = lane % LPR;
    uint32_t raddr[NPP], rsw[NPP];
#pragma unroll
    for (int n = 0; n < NPP; ++n) {
        const int row = n * 32 + l31;
        raddr[n] = row * BR;
        rsw[n] = (row >> SWS) & SWM;
    }
    u32x4 st[NI];
    typename E::v8 wf[NSW][NT3], xf[2][NPP];  // token fragments: this k-step's and the next one's
#define SFS_GLOAD(c_)                                                                                                     \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                                      \
        int tok_ = pan[(i * RPI) >> 5] * 32 + ((i * RPI) & 31) + rsub;                                                    \
        tok_ = tok_ < N ? tok_ : N - 1;                                                                                   \
        st[i] = *reinterpret_cast<const u32x4*>(xb + (uint32_t)tok_ * (C * 2) + piece * 16 + (c_) * BR);                  \
    }
#define SFS_WRITE()                                                                                                       \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) {                                                                      \
        const int row_ = i * RPI + rsub;                                                                                  \
        *reinterpret_cast<u32x4*>(stg + row_ * BR + ((piece ^ ((row_ >> SWS) & SWM)) << 4)) = st[i];                      \
    }
#define SFS_READ(s_, kl_)                                                                                                 \
    _Pragma("unroll") for (int n = 0; n < NPP; ++n)                                                                       \
        xf[s_][n] = as_v8<DT>(*reinterpret_cast<const uint4*>(stg + raddr[n] + ((((uint32_t)((kl_) * 2 + half)) ^ rsw[n]) << 4)));
#define SFS_LDW(i_, kk_) _Pragma("unroll") for (int j = 0; j < NT3; ++j) wf[i_][j] = __builtin_bit_cast(typename E::v8, *(sf_gptr16)(wb + (j * KC + (kk_)) * 1024 + loff));
    SFS_GLOAD(0);
#pragma unroll
    for (int i = 0; i < NSW; ++i) { SFS_LDW(i, i); }
    SFS_WRITE();
    if (NCH > 1) { SFS_GLOAD(1); }
    SFS_READ(0, 0);
#pragma unroll
    for (int n = 0; n < NPP; ++n) {
        shift[n] = SF_STAT == 1 ? 0.f : half_lo((float)xf[0][n][0]);
        ssum[n] = 0.f;
        sq[n] = 0.f;
    }
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int i = 0; i < KCH; ++i) {
            if (i + 1 < KCH) { SFS_READ((i + 1) & 1, i + 1); }
#pragma unroll
            for (int n = 0; n < NPP; ++n) {
#pragma unroll
                for (int j = 0; j < NT3; ++j) acc[n][j] = E::mfma32(wf[i % NSW][j], xf[i & 1][n], acc[n][j]);
                if (SF_STAT == 1) {
#pragma unroll
                    for (int w_ = 0; w_ < 4; ++w_) {
                        const uint32_t pr_ = __builtin_bit_cast(u32x4, xf[i & 1][n])[w_];
                        ssum[n] = sf_dot2<DT>(pr_, sf_ones2<DT>(), ssum[n]);
                        sq[n] = sf_dot2<DT>(pr_, pr_, sq[n]);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d_ = (float)xf[i & 1][n][e] - shift[n];
                        ssum[n] += d_;
                        sq[n] = __builtin_fmaf(d_, d_, sq[n]);
                    }
                }
            }
            const int kn = c * KCH + i + NSW;  // (past the end: the last k-step again, never used)
            SFS_LDW(i % NSW, kn < KC ? kn : KC - 1);
        }
        if (c + 1 < NCH) {  // the next chunk: registers -> the window (every read of this chunk is issued), the one after it -> registers
            SFS_WRITE();
            if (c + 2 < NCH) { SFS_GLOAD(c + 2); }
            SFS_READ(0, 0);
        }
    }
#undef SFS_GLOAD
#undef SFS_WRITE
#undef SFS_READ
#undef SFS_LDW
}

// (two waves per SIMD by contract: at d = 48 the four-wave workgroups need TWO per CU -- round 6 found the kernel at 256 + 44 registers, i.e. one workgroup per
//  CU and its 512 workgroups in two rounds of 22 us; SF_OCC = 1 re-creates that build)
template <int DT, int D, int KC, int NW, int NPP, int NSET, bool STAGED>
__global__ __launch_bounds__(NW * 64, SF_OCC) void sattn_fused_kernel(SfP p) {
    using E = ET<DT>;
    using Y = Lay<D>;
    // the head's q | k | v rows are packed DENSELY: 3 d virtual rows in ceil(3 d / 32) row tiles (d = 48: 4.5 -> 5 tiles, not 3 x 2); an 8-row
    // group never straddles two of the three (d % 8 == 0)
    constexpr int C = KC * 16, D8 = D / 8, NT3 = (3 * D + 31) / 32, KCD = D / 16, NPW = 2;  // NPW: query panels a wave attends at once
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int b = (seq / p.H) * 8 + xcd, h = seq % p.H;  // (all heads of a sample on one XCD: they share its rows)
    if (b >= p.B) return;
    SF_STAMP(0);
    const int N = p.N;
    const int npan = (N + 31) >> 5, ntiles = (N + KT - 1) / KT, nfull = N / KT;
    const int rounds = (npan + NPW * NW - 1) / (NPW * NW);
    float* const csbb = reinterpret_cast<float*>(smem + ntiles * Y::BUF);  // [2][NT3 * 32]
    // the head's colsum / bias vectors -> LDS; the last key tile is cleared (its unwritten V^T columns / rows meet zero probabilities).
    // STAGED: neither is needed before the first epilogue, which sits behind a workgroup barrier anyway -- the vectors are requested here and parked in LDS behind
    // the first projection (their HBM round trip under its loads instead of in front of them: 1.5 us per workgroup), the zero paddings are written behind the
    // last round's fence (the staging windows would overwrite them)
    constexpr int NCSB = (2 * NT3 * 32 + NW * 64 - 1) / (NW * 64);
    float csreg[NCSB];
#pragma unroll
    for (int k = 0; k < NCSB; ++k) {
        const int i = tid + k * NW * 64;
        csreg[k] = i < 2 * NT3 * 32 ? p.csbb[(int64_t)h * 2 * NT3 * 32 + i] : 0.f;
    }
    if constexpr (!STAGED) {
#pragma unroll
        for (int k = 0; k < NCSB; ++k)
            if (tid + k * NW * 64 < 2 * NT3 * 32) csbb[tid + k * NW * 64] = csreg[k];
        for (int i = tid; i < Y::BUF / 4; i += NW * 64) reinterpret_cast<uint32_t*>(smem + (ntiles - 1) * Y::BUF)[i] = 0u;
        if (Y::VROWS > D) {  // (d = 48: rows 48 .. 63 of every V^T tile feed discarded output rows, but must be finite)
            for (int t = 0; t < ntiles - 1; ++t)
                for (int i = tid; i < (Y::VROWS - D) * Y::VROW / 4; i += NW * 64) reinterpret_cast<uint32_t*>(smem + t * Y::BUF + Y::K_BYTES + D * Y::VROW)[i] = 0u;
        }
        __syncthreads();
    }
    SF_STAMP(1);
    const sf_gptr wb = sf_sgpr_ptr(p.w + (int64_t)h * NT3 * KC * 1024);
    const uint32_t loff = (uint32_t)lane * 16u;
    const uint8_t* const xb = p.x + (int64_t)b * N * C * 2;
    // rounds of (NPW x NW) panels a workgroup needs: 1024 tokens / 8 waves = 2; the 384-wide level's <= 256 tokens / 4 waves = 1 (apad_self_attention_fused's envelope)
    constexpr int MAXR = D == 32 ? 2 : 1;
    typename E::v8 qkeep[MAXR][NPW][KCD];  // [round][panel of the pair][k-step]: this wave's q fragments, projection -> key loop
    // ---- 1. projection: K, V^T -> LDS tiles; Q -> registers ----
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r >= rounds) break;
#pragma unroll
        for (int g0 = 0; g0 < NPW; g0 += NPP) {
            int pan[NPP];
            const uint8_t* xrow[NPP];
            bool act = false;
#pragma unroll
            for (int n = 0; n < NPP; ++n) {
                pan[n] = (r * NPW + g0 + n) * NW + wave;
                int tok = pan[n] * 32 + l31;
                tok = tok < N ? tok : N - 1;
                xrow[n] = xb + ((int64_t)tok * C + half * 8) * 2;
                if (SF_ABL & 64) xrow[n] = xb + ((int64_t)min(pan[n] * 32 + half, N - 32) * C + l31 * 8) * 2;  // (timing only: two whole rows per load instruction)
                act = act || pan[n] < npan;
            }
            // STAGED (sf_project_stg): every wave's 8 KB window at the END of the key-tile region.  sf_go launches this instantiation only where no EARLIER round
            // has written tiles there (1000 tokens: round 0 fills tiles 0 .. 7, the windows sit in tiles 9 .. 15); a round whose own epilogue writes into the
            // region is fenced from it by a workgroup barrier, behind which the zero paddings the windows overwrote are restored (disjoint from what the epilogues write)
            constexpr int STG_KCH = D == 32 ? 4 : 2, STG = NPP * 32 * STG_KCH * 32, TILES_R = NPW * NW * 32 / KT;
            static_assert(!STAGED || NPP == NPW, "a wave stages the 64 rows of both of its panels");
            const int stg_off = ntiles * Y::BUF - NW * STG;
            const bool fence = STAGED && ((r + 1) * TILES_R < ntiles ? (r + 1) * TILES_R : ntiles) * Y::BUF > stg_off;  // (workgroup-uniform)
            f32x16 acc[NPP][NT3];
#pragma unroll
            for (int n = 0; n < NPP; ++n)
#pragma unroll
                for (int j = 0; j < NT3; ++j)
#pragma unroll
                    for (int q_ = 0; q_ < 16; ++q_) acc[n][j][q_] = 0.f;
            float ssum[NPP], sq[NPP], shift[NPP];
            if (SF_ABL & 2) {
#pragma unroll
                for (int n = 0; n < NPP; ++n) ssum[n] = sq[n] = shift[n] = 1.f;
            } else if (act) {  // (wave-uniform)
                if constexpr (STAGED) sf_project_stg<DT, NT3, NPP, KC, STG_KCH>(wb, loff, xb, pan, N, smem + stg_off + wave * STG, lane, acc, ssum, sq, shift);
                else sf_project<DT, NT3, NPP, KC, NSET>(wb, loff, xrow, acc, ssum, sq, shift);
            }
            SF_STAMP(2 + 2 * r);
            if (STAGED && r == 0) {
#pragma unroll
                for (int k = 0; k < NCSB; ++k)
                    if (tid + k * NW * 64 < 2 * NT3 * 32) csbb[tid + k * NW * 64] = csreg[k];
            }
            if (fence || (STAGED && r == 0)) __syncthreads();
            if (fence) {
                if (Y::VROWS > D) {
                    for (int t = 0; t < ntiles; ++t)
                        for (int i = tid; i < (Y::VROWS - D) * Y::VROW / 4; i += NW * 64) reinterpret_cast<uint32_t*>(smem + t * Y::BUF + Y::K_BYTES + D * Y::VROW)[i] = 0u;
                }
                if (npan & 1) {  // keys 32 .. 63 of the last tile belong to no panel: K rows 32 .. 63, bytes 64 .. 127 of every V^T row
                    uint8_t* const lt = smem + (ntiles - 1) * Y::BUF;
                    for (int i = tid; i < 32 * Y::KROW / 4; i += NW * 64) reinterpret_cast<uint32_t*>(lt + 32 * Y::KROW)[i] = 0u;
                    for (int i = tid; i < Y::VROWS * 16; i += NW * 64) reinterpret_cast<uint32_t*>(lt + Y::K_BYTES + (i >> 4) * Y::VROW + 64)[i & 15] = 0u;
                }
            }
            if (act) {
#pragma unroll
            for (int n = 0; n < NPP; ++n) {
                const int key = pan[n] * 32 + l31;
                const bool ok = key < N;
                const float s1 = half_sum(ssum[n]), s2 = half_sum(sq[n]);
                const float md = s1 * (1.0f / C), mean = shift[n] + md;
                const float var = fmaxf(s2 * (1.0f / C) - md * md, 0.f);
                const float rstd = (SF_ABL & 4) ? (ok ? 1.f : 0.f) : (ok ? rsqrtf(var + p.eps) : 0.f), nmr = (SF_ABL & 4) ? 0.f : -mean * rstd, okf = ok ? 1.f : 0.f;
                uint8_t* const kt = smem + (key >> 6) * Y::BUF;
                const int krow = key & 63;
                const int kpos = (krow & ~12) | ((krow & 4) << 1) | ((krow & 8) >> 1);  // position of the key inside the tile's V^T rows (Lay::VROW: bits 2, 3 swapped)
#pragma unroll
                for (int j = 0; j < NT3; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int vr8 = 4 * j + g;                    // 8-row group of the head's virtual rows
                        if (vr8 >= 3 * D8) continue;                  // (the padding rows of the last tile)
                        const int which = vr8 / D8, G = vr8 % D8;     // 0 q, 1 k, 2 v; 8-feature group inside the head
                        const int fl = 8 * g + 4 * half;              // first of this lane's 4 consecutive rows inside the tile
                        const float4 cs = *reinterpret_cast<const float4*>(csbb + j * 32 + fl);
                        const float4 bb = *reinterpret_cast<const float4*>(csbb + NT3 * 32 + j * 32 + fl);
                        float y[4];
                        y[0] = __builtin_fmaf(acc[n][j][4 * g + 0], rstd, __builtin_fmaf(cs.x, nmr, bb.x * okf));
                        y[1] = __builtin_fmaf(acc[n][j][4 * g + 1], rstd, __builtin_fmaf(cs.y, nmr, bb.y * okf));
                        y[2] = __builtin_fmaf(acc[n][j][4 * g + 2], rstd, __builtin_fmaf(cs.z, nmr, bb.z * okf));
                        y[3] = __builtin_fmaf(acc[n][j][4 * g + 3], rstd, __builtin_fmaf(cs.w, nmr, bb.w * okf));
                        if (which == 0) {  // q: registers 8 cc .. 8 cc + 7 of the tile are the B-operand fragment of k-step 2 jt + cc
#pragma unroll
                            for (int e = 0; e < 4; ++e) qkeep[r][g0 + n][G >> 1][4 * (G & 1) + e] = (typename E::elem)y[e];
                        } else if (which == 1) {  // k: [key][d] with d permuted inside each 16-wide k-step exactly as q's fragments hold it
                            typename E::v4 kv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) kv[e] = (typename E::elem)y[e];
                            if (pan[n] < npan)
                                *reinterpret_cast<uint2*>(kt + krow * Y::KROW + (16 * (G >> 1) + 8 * half + 4 * (G & 1)) * 2) = __builtin_bit_cast(uint2, kv);
                        } else {  // v: transposed, V^T[d][key] (natural key order: the P.V product reads it with the C-layout key permutation)
                            if (pan[n] < npan) {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    *reinterpret_cast<typename E::elem*>(kt + Y::K_BYTES + (8 * G + 4 * half + e) * Y::VROW + kpos * 2) = (typename E::elem)y[e];
                            }
                        }
                    }
                }
            }
            }
            SF_STAMP(3 + 2 * r);
        }
    }
    __syncthreads();
    SF_STAMP(6);
    // ---- 2. attention over the resident tiles: two query panels per wave and round ----
#pragma unroll
    for (int r = 0; r < MAXR; ++r) {
        if (r >= rounds) break;
        const int pa = (r * NPW) * NW + wave, pb = pa + NW;
        if (pa >= npan || (SF_ABL & 1)) continue;  // (wave-uniform; pb >= npan: its lanes carry the clamped last row and are not stored)
        f32x16 o[2][Y::DT_TILES];
        f32x2 osum[2];
        float m[2];
        f32x16 mi[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            osum[qt] = (f32x2){0.f, 0.f};
            m[qt] = NEG_BIG;
#pragma unroll
            for (int q_ = 0; q_ < 16; ++q_) mi[qt][q_] = -NEG_BIG;  // exp2(s + 1e30) = inf: the first tile takes the classic path
#pragma unroll
            for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
                for (int q_ = 0; q_ < 16; ++q_) o[qt][dt][q_] = 0.f;
        }
        int t = 0;
#pragma unroll 1
        for (; t < nfull; ++t) tile_compute2<DT, D, false, SF_DIRECT(D)>(smem + t * Y::BUF, t * KT, N, 1.0f, qkeep[r], o, osum, m, mi, l31, half);
        if (t < ntiles) tile_compute2<DT, D, true, SF_DIRECT(D)>(smem + t * Y::BUF, t * KT, N, 1.0f, qkeep[r], o, osum, m, mi, l31, half);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int q = (qt == 0 ? pa : pb) * 32 + l31;
            const float inv = 1.0f / half_sum(osum[qt][0] + osum[qt][1]);
            if (q < N) {
                uint8_t* const ob = p.out + (((int64_t)b * N + q) * C + h * D) * 2;
#pragma unroll
                for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int dcol = dt * 32 + 8 * g + 4 * half;
                        if (dcol < D) {
                            typename E::v4 pk;
#pragma unroll
                            for (int e = 0; e < 4; ++e) pk[e] = (typename E::elem)(o[qt][dt][g * 4 + e] * inv);
                            *reinterpret_cast<uint2*>(ob + dcol * 2) = __builtin_bit_cast(uint2, pk);
                        }
                    }
            }
        }
        SF_STAMP(7 + r);
    }
    SF_STAMP(9);
}
#ifdef SF_TRACE
extern "C" int apad_sf_trace_read(void* dst, int bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(sf_trace_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

template <int DT, int D, int KC, int NW, int NPP, int NSET, bool STAGED = false> int sf_go(const SfP& p, hipStream_t s) {
    using Y = Lay<D>;
    constexpr int NT3 = (3 * D + 31) / 32;
    const int ntiles = (p.N + KT - 1) / KT;
    if constexpr (!STAGED && SF_STAGE(D) && NPP == 2) {
        // the LDS-staged projection (sf_project_stg) where every wave's 8 KB window fits behind the tiles the earlier rounds fill
        const int npan = (p.N + 31) / 32, rounds = (npan + 2 * NW - 1) / (2 * NW), stg_off = ntiles * Y::BUF - NW * NPP * 32 * (D == 32 ? 128 : 64);
        if (stg_off >= (rounds - 1) * (2 * NW * 32 / KT) * Y::BUF) return sf_go<DT, D, KC, NW, NPP, NSET, true>(p, s);
    }
    const int lds = ntiles * Y::BUF + 2 * NT3 * 32 * 4;
    auto kern = sattn_fused_kernel<DT, D, KC, NW, NPP, NSET, STAGED>;
    static unsigned devs = 0;
    constexpr int MAXT = D == 32 ? 16 : 4;  // key tiles of the largest routed sequence (1024 / 256 tokens)
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), MAXT * Y::BUF + 2 * NT3 * 32 * 4, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(((p.B + 7) / 8) * 8 * p.H)), dim3(NW * 64), lds, s, p);
    return apad_check_launch("apad_self_attention_fused");
}

// ---------------------------------------------------------------------------------------------------------------------
// Short-segment variant: every softmax segment has at most 64 keys (the adapter's decoupled cross-attention at
// La <= 64 -- 8 text + 32 audio tokens in the style_transfer preset --, the 16-token T5 cross-attention).  Such a launch
// is bound by reading Q and writing O; the staged kernel above spends it on LDS staging and four workgroup barriers per
// 128 queries.  Here a wave is independent: K fragments (A operand rows = keys) and V^T fragments come straight from
// global memory (a few KB per (batch, head), L2-resident), one masked tile per segment, no online rescale, row sums by a
// half-wave exchange; only the output transpose goes through a per-wave LDS scratch.
template <int DT, int D>
__device__ __forceinline__ void short_segment(const uint8_t* kbase, int64_t k_sl, const uint8_t* vbase, int L, int Lpad, const float* bias,
                                              float c, const typename ET<DT>::v8* qf, f32x16* o, float& inv_den, int l31, int half) {
    using E = ET<DT>;
    using Y = Lay<D>;
    constexpr int KC = D / 16;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nsub = L > 32 ? 2 : 1;
    f32x16 s[2];
    s[0] = s[1] = zero16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nsub) break;
        const int key = u * 32 + l31;
        const uint8_t* kp = kbase + ((int64_t)(key < L ? key : L - 1) * k_sl + half * 8) * 2;  // rows past L: masked below
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) {
            typename E::v8 kf = as_v8<DT>(*reinterpret_cast<const uint4*>(kp + cc * 32));
            s[u] = E::mfma32(kf, qf[cc], cc == 0 ? zero16 : s[u]);
        }
    }
    float tmax = NEG_BIG;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nsub) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = s[u][r] * c;
            if (bias) v += bias[key < L ? key : L - 1] * LOG2E;
            v = key < L ? v : NEG_BIG;
            s[u][r] = v;
            tmax = fmaxf(tmax, v);
        }
    }
    tmax = half_max(tmax);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nsub) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // the denominator sums the probabilities as the P.V MFMA sees them (rounded to the storage type)
            const float e = (float)(typename E::elem)__builtin_amdgcn_exp2f(s[u][r] - tmax);
            s[u][r] = e;
            sum += e;
        }
    }
    sum = half_sum(sum);
    inv_den = 1.0f / sum;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st >= 2 * nsub) break;
        typename E::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[st >> 1][(st & 1) * 8 + j];
        const int kcol = st * 16 + 4 * half;
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt) {
            const int d = dt * 32 + l31;
            uint2 v0 = make_uint2(0u, 0u), v1 = make_uint2(0u, 0u);
            if (d < D) {  // kcol + 11 < Lpad: Lpad is a multiple of 32 covering every sub-tile that is visited
                const uint8_t* vp = vbase + ((int64_t)d * Lpad + kcol) * 2;
                v0 = *reinterpret_cast<const uint2*>(vp);
                v1 = *reinterpret_cast<const uint2*>(vp + 16);
            }
            o[dt] = E::mfma32(as_v8<DT>(make_uint4(v0.x, v0.y, v1.x, v1.y)), pf, o[dt]);
        }
    }
}

template <int DT, int D, bool DUAL>
__global__ __launch_bounds__(256) void attn_short_kernel(AttnP p) {
    // (a one-workgroup-per-32-queries-x-all-heads shape, meant to consume whole 2*C-byte rows while their lines are hot,
    //  measured slower: 38 vs 28 us at the 1000-token level -- the heads serialise inside a wave)
    using E = ET<DT>;
    using Y = Lay<D>;
    constexpr int KC = D / 16;
    constexpr int OROW = D * 2 + 8;
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * 32 * OROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, h = bh % p.H, b = bh / p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    if (q0 >= p.N) return;  // waves are independent: no barrier below
    int qi = q0 + l31;
    qi = qi < p.N ? qi : p.N - 1;
    typename E::v8 qf[KC];
    const uint8_t* qp = p.q + ((int64_t)b * p.q_sb + (int64_t)qi * p.q_sn + h * D + half * 8) * 2;
#pragma unroll
    for (int cc = 0; cc < KC; ++cc) qf[cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(qp + cc * 32));
    f32x16 o[Y::DT_TILES];
#pragma unroll
    for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float inv = 1.f;
    {
        const int bk = b / p.kvdiv;
        const uint8_t* kbase = p.k + ((int64_t)bk * p.k_sb + h * D) * 2;
        const uint8_t* vbase = p.vt + ((int64_t)bk * p.vt_sb + (int64_t)h * D * p.Lpad) * 2;
        const float* bias = p.key_bias ? p.key_bias + (int64_t)b * p.L : nullptr;
        short_segment<DT, D>(kbase, p.k_sl, vbase, p.L, p.Lpad, bias, p.scale_log2, qf, o, inv, l31, half);
    }
#pragma unroll
    for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= inv;
    if (DUAL) {
        f32x16 o2[Y::DT_TILES];
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o2[dt][r] = 0.f;
        float inv2 = 1.f;
        const int bk = b / p.kvdiv2;
        const uint8_t* kbase = p.k2 + ((int64_t)bk * p.k2_sb + h * D) * 2;
        const uint8_t* vbase = p.vt2 + ((int64_t)bk * p.vt2_sb + (int64_t)h * D * p.Lpad2) * 2;
        short_segment<DT, D>(kbase, p.k2_sl, vbase, p.L2, p.Lpad2, nullptr, p.scale_log2, qf, o2, inv2, l31, half);
        // the un-fused reference rounds each branch, and scale * audio, to the storage type before the add
#pragma unroll
        for (int dt = 0; dt < Y::DT_TILES; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float t = (float)(typename E::elem)o[dt][r];
                const float a = (float)(typename E::elem)(o2[dt][r] * inv2);
                o[dt][r] = t + (float)(typename E::elem)(p.scale2 * a);
            }
    }
    uint8_t* scr = smem + wave * (32 * OROW);
#pragma unroll
    for (int dt = 0; dt < Y::DT_TILES; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dcol = dt * 32 + 8 * g + 4 * half;
            if (dcol < D) {
                typename E::v4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)o[dt][g * 4 + j];
                *reinterpret_cast<uint2*>(scr + l31 * OROW + dcol * 2) = __builtin_bit_cast(uint2, pk);
            }
        }
    }
    constexpr int CPR = D / 8;
    uint8_t* ob = p.out + ((int64_t)b * p.o_sb + h * D) * 2;
    for (int idx = lane; idx < 32 * CPR; idx += 64) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const int q = q0 + row;
        if (q < p.N) {
            const uint2 lo = *reinterpret_cast<const uint2*>(scr + row * OROW + ch * 16);
            const uint2 hi = *reinterpret_cast<const uint2*>(scr + row * OROW + ch * 16 + 8);
            *reinterpret_cast<uint4*>(ob + ((int64_t)q * p.o_sn + ch * 8) * 2) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
    }
}

template <int DT, int D> int launch_d(const AttnP& p, bool dual, dim3 grid, hipStream_t s) {
    constexpr bool no_short = false;
    if (!no_short && p.L <= 64 && (!dual || p.L2 <= 64) && p.lse == nullptr) {
        dim3 g2((unsigned)((p.N + 127) / 128), (unsigned)(p.B * p.H));
        if (dual)
            hipLaunchKernelGGL((attn_short_kernel<DT, D, true>), g2, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((attn_short_kernel<DT, D, false>), g2, dim3(256), 0, s, p);
        return apad_check_launch("apad_attention");
    }
    if constexpr (D == 32 || D == 48 || D == 64) {
        // long single-segment launches without a key bias (the UNet's self-attention): two query tiles per wave
        constexpr int two_q = 1;
        // (A/B knob; round 3: 512 -> 200, i.e. the 252-token level's d = 48 self-attention too: step 44.87 -> 44.74 ms; in round 2,
        //  before the batched fragment reads, it measured slower there; the gain is 0.1 ms)
        constexpr int two_q_min = 200;
        if (two_q && !dual && p.key_bias == nullptr && p.N >= two_q_min && p.L >= (two_q_min < 256 ? two_q_min : 256) && (D == 32 || D == 48 || two_q > 1)) {
            if constexpr (D == 32) {
                constexpr int nw8 = 0;  // off: step 49.68 -> 50.12 ms (the 8-wave barrier costs more than the halved staging saves)
                if (nw8) {
                    dim3 g8((unsigned)(((p.N + 511) / 512) * 8 * ((p.B + 7) / 8) * p.H));
                    hipLaunchKernelGGL((attn2q_kernel<DT, D, 8>), g8, dim3(512), 0, s, p);
                    return apad_check_launch("apad_attention");
                }
            }
            dim3 g2((unsigned)(((p.N + 255) / 256) * 8 * ((p.B + 7) / 8) * p.H));  // (sample-major XCD order: attn2q_body)
            if constexpr (D == 32) {
                // pre-scaled q: the softmax without per-score max / scale instructions.  d = 48 (the
                // 252-token level) keeps the classic two-tile form: its direct form needs 256 VGPRs and measured slower in-step
                // (44.82 vs 44.69 ms)
                constexpr int direct = 1;
                if (direct && p.prescaled) {
                    hipLaunchKernelGGL((attn2q_kernel<DT, D, 4, true>), g2, dim3(256), 0, s, p);
                    return apad_check_launch("apad_attention");
                }
            }
            hipLaunchKernelGGL((attn2q_kernel<DT, D, 4>), g2, dim3(256), 0, s, p);
            return apad_check_launch("apad_attention");
        }
    }
    if (dual)
        hipLaunchKernelGGL((attn_kernel<DT, D, true>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((attn_kernel<DT, D, false>), grid, dim3(256), 0, s, p);
    return apad_check_launch("apad_attention");
}

template <int DT> int launch_dt(const AttnP& p, int D, bool dual, dim3 grid, hipStream_t s) {
    switch (D) {
        case 16: return launch_d<DT, 16>(p, dual, grid, s);
        case 32: return launch_d<DT, 32>(p, dual, grid, s);
        case 48: return launch_d<DT, 48>(p, dual, grid, s);
        case 64: return launch_d<DT, 64>(p, dual, grid, s);
        case 80: return launch_d<DT, 80>(p, dual, grid, s);
        case 96: return launch_d<DT, 96>(p, dual, grid, s);
        case 128: return launch_d<DT, 128>(p, dual, grid, s);
    }
    apad_set_error("apad_attention: head dim %d not supported (16,32,48,64,80,96,128)", D);
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// apad_cross_attention_rows: the fused cross-attention sub-layer (apad_fused_cross_attention's arithmetic) for the levels whose width
// does not fit xattn.hip's weight-stationary registers -- C = 384 (252 tokens per sample, d = 48):
//     out = x + to_out( A(q, K1, V1, bias) [+ scale2 * A(q, K2, V2)] ) + b_out,   q = to_q(LayerNorm(x))
// One 256-thread workgroup owns 64 tokens of ONE sample (4 tiles per sample at 252 tokens: 256 workgroups for the CFG batch of 64 = one
// per CU).  The tokens stay in LDS from the first read to the last write -- two [64][C] tiles, rows padded to an odd number of 16-byte
// slots -- and the three kernels of the un-fused chain become four phases separated by workgroup barriers:
//   1. LayerNorm: x -> X tile (8 lanes per row, statistics in registers, two passes)
//   2. q^T = Wq . X^T -> Q tile.  Wave w owns output features 96 w .. 96 w + 95 (3 MFMA row tiles) for BOTH 32-token panels: every weight
//      fragment is read ONCE per workgroup, straight from L2 into registers (fragment-major packing: one contiguous KB per wave-load, two
//      k-steps ahead); the token fragments come from the X tile.  C layout = (lane: token, registers: 4 consecutive features) -> 8-byte
//      LDS stores into the row-major Q tile
//   3. attention: wave w runs heads 2 w, 2 w + 1 on both panels with attn_short_kernel's segment routine (K rows and V^T rows of the
//      hoisted sets straight from L2, <= 64 keys per segment, each branch rounded before the blend); q fragments from the Q tile, the
//      heads' outputs into the X tile (the normalised tokens are dead)
//   4. out^T = Wo . O^T (+ bias) -> Q tile (rounded like the chain's to_out), then one coalesced pass adds the residual x and stores.
// HBM traffic per launch: x once in, out once out (the chain: six activation passes).
#ifndef XR_ABL
#define XR_ABL 0  // timing ablations (tools/ab_build.sh; results are wrong): 1 = weight fragments loaded once, 2 = no attention phase, 4 = no projections, 8 = no LayerNorm arithmetic
#endif
struct XrP {
    const uint8_t* x;
    const uint8_t* gamma;
    const uint8_t* beta;
    const uint8_t* wq;  // packed: [C / 32 row tiles][C / 16 k-steps][64 lanes][8]
    const uint8_t* wo;
    const uint8_t* bo;
    const uint8_t* k1;  // vt1 == nullptr: apad_rows_pack_kv's fragment packing of the segment (KvPacked); else apad_attention's layout (KvRaw)
    const uint8_t* vt1;
    const float* bias1;
    const uint8_t* k2;
    const uint8_t* vt2;
    uint8_t* out;
    int32_t B, N, L1, Lpad1, L2, Lpad2, tiles_per_sample;
    float eps, scale_log2, scale2;
};

#include "short_seg.h"

// the two projections of xattn_rows_kernel: dst^T[feature][token] = W . src^T (+ bias), wave w = features 32 NT w .. (+ 32 NT), both
// 32-token panels; weight fragments from L2 (packed), three register sets rotating two k-steps ahead
template <int DT, int NT>
__device__ __forceinline__ void xr_ldw(typename ET<DT>::v8 (&wf)[NT], const uint8_t* wl, int KS, int kk) {
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = as_v8<DT>(*reinterpret_cast<const uint4*>(wl + ((int64_t)j * KS + kk) * 1024));
}
template <int DT, int NT, int ROWB, int PW>
__device__ __forceinline__ void xr_step(f32x16 (&acc)[NT][PW], const typename ET<DT>::v8 (&wf)[NT], const uint8_t* sl, int kk) {
    using E = ET<DT>;
    typename E::v8 t[PW];
#pragma unroll
    for (int mt = 0; mt < PW; ++mt) t[mt] = as_v8<DT>(*reinterpret_cast<const uint4*>(sl + mt * 32 * ROWB + kk * 32));
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int mt = 0; mt < PW; ++mt) acc[j][mt] = E::mfma32(wf[j], t[mt], acc[j][mt]);
}
template <int DT, int C, int PW, int NSET>
__device__ __forceinline__ void xr_project(const uint8_t* wpk, const uint8_t* src, uint8_t* dst, const uint8_t* bias, int wave, int lane) {
    // PW = token panels per wave: 2 -> 4 waves (every weight fragment read once per workgroup), 1 -> 8 waves (waves w and w + 4 share
    // the fragments through the CU's vector cache; twice the loads in flight, two waves per SIMD)
    using E = ET<DT>;
    constexpr int KS = C / 16, NT = C / 32 / 4, ROWB = C * 2 + 16;
    const int p0 = PW == 1 ? (wave >> 2) : 0;
    wave &= 3;
    src += p0 * 32 * ROWB;
    dst += p0 * 32 * ROWB;
    static_assert(KS % NSET == 0, "the register sets of weight fragments rotate over the k-steps");
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[NT][PW];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int mt = 0; mt < PW; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][mt][r] = 0.f;
    const uint8_t* wl = wpk + ((int64_t)(wave * NT) * KS) * 1024 + lane * 16;
    const uint8_t* sl = src + l31 * ROWB + half * 16;
    typename E::v8 wf[NSET][NT];  // NSET register sets of weight fragments: the fragment of k-step kk + NSET is requested when kk is consumed
#pragma unroll
    for (int i = 0; i < NSET; ++i) xr_ldw<DT, NT>(wf[i], wl, KS, i);
#pragma unroll 1
    for (int kk = 0; kk < KS; kk += NSET) {
#pragma unroll
        for (int i = 0; i < NSET; ++i) {
            xr_step<DT, NT, ROWB, PW>(acc, wf[i], sl, kk + i);
            if (!(XR_ABL & 1) && kk + i + NSET < KS) xr_ldw<DT, NT>(wf[i], wl, KS, kk + i + NSET);
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int f0 = (wave * NT + j) * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = f0 + 8 * g + 4 * half;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias != nullptr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = ld_elem<DT>(bias, f + e);
            }
#pragma unroll
            for (int mt = 0; mt < PW; ++mt) {
                typename E::v4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (typename E::elem)(acc[j][mt][4 * g + e] + bv[e]);
                *reinterpret_cast<uint2*>(dst + (mt * 32 + l31) * ROWB + f * 2) = __builtin_bit_cast(uint2, y);
            }
        }
    }
}

template <int DT, int C, int NS1, int NS2, int NW, int NSET>
__global__ __launch_bounds__(NW * 64) void xattn_rows_kernel(XrP p) {
    constexpr bool DUAL = NS2 > 0;
    // one 32-token panel per wave quartet: 8 waves = 64 tokens (C = 384), 4 waves = 32 tokens (C = 640: two 64-token tiles would not fit the LDS)
    constexpr int PW = 1, NTH = NW * 64, XR_TM = NW * 8;
    static_assert(NW == 4 || NW == 8, "one or two wave quartets");
    using E = ET<DT>;
    constexpr int H = 8, D = C / H, KC = D / 16, DTT = (D + 31) / 32;
    constexpr int ROWB = C * 2 + 16;  // 49 (C = 384) sixteen-byte slots: odd -> conflict-free fragment reads over 32 rows
    constexpr int CH = C / 64;        // 16-byte chunks per lane in the LayerNorm pass (8 lanes per row)
    static_assert(C % 128 == 0 && D % 16 == 0, "4 waves x whole row tiles; whole k-steps per head");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const X = smem;
    uint8_t* const Q = smem + XR_TM * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // XCD-aware (speed only; round 6): workgroup i runs on XCD i % 8, so the tiles of ONE sample -- which read the same key / value sets, 0.9 MB per sample at
    // 512 audio keys -- are given to one XCD's L2 instead of four (the plain order had every set fetched through the fabric four times)
    int bid = blockIdx.x;
    {
        const int per = gridDim.x >> 3;
        if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
    }
    const int b = bid / p.tiles_per_sample, row0 = (bid - b * p.tiles_per_sample) * XR_TM;
    const int nrows = p.N - row0 < XR_TM ? p.N - row0 : XR_TM;
    const uint8_t* const xb = p.x + ((int64_t)b * p.N + row0) * C * 2;

    // ---- 1. LayerNorm -> X ----
    {
        const int sub = tid & 7;
#pragma unroll
        for (int it = 0; it < XR_TM / (NTH / 8); ++it) {
            const int row = it * (NTH / 8) + (tid >> 3);
            float v[CH][8];
            const bool ok = row < nrows;
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                uint4 u = make_uint4(0u, 0u, 0u, 0u);
                if (ok) u = *reinterpret_cast<const uint4*>(xb + ((int64_t)row * C + (sub + 8 * i) * 8) * 2);
                unpack8<DT>(u, v[i]);
            }
            if (p.gamma != nullptr && !(XR_ABL & 8)) {
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
    }
    __syncthreads();

    // ---- 2. q = to_q(X) -> Q ----
    if (!(XR_ABL & 4)) xr_project<DT, C, PW, NSET>(p.wq, X, Q, nullptr, wave, lane);

    // ---- 3. attention, heads 2 w and 2 w + 1: Q -> X.  All K / V^T fragments of a head (both segments) are requested before any of its
    //         arithmetic, and the first head's before the barrier: one exposed L2 round trip per wave ----
    constexpr bool BIG2 = NS2 > 2;  // the second segment's fragments are requested as they are used (short_segment_ns)
    constexpr bool LONG2 = NS2 > 4;  // ... in 64-key chunks with a running maximum / sum (long_segment: 129 .. 512 audio keys)
    constexpr int NSB = (DUAL && !BIG2) ? NS2 : 1;
    ShortFr<DT, D, NS1> f1;
    ShortFr<DT, D, NSB> f2;
    // (two sub-tiles in both segments: both fragment sets at once do not fit the 256 registers of two waves per SIMD -- that form keeps
    //  the load-as-you-go segment routine)
    constexpr bool SPLITF = DUAL && !BIG2 && NS1 + NS2 > 3;
    // the fragment sources of a head's two segments: fragment-packed sets (round 6) or apad_attention's row-major / transposed tensors
    const bool pk1 = p.vt1 == nullptr, pk2 = DUAL && p.vt2 == nullptr;
    const int64_t hb1 = kv_packed_head_bytes(D, p.L1), hb2 = kv_packed_head_bytes(D, p.L2);
#define XR_RAW1(h_) KvRaw<DT, D>{p.k1 + ((int64_t)b * p.L1 * C + (h_) * D) * 2, C, p.vt1 + ((int64_t)(b * H + (h_)) * D * p.Lpad1) * 2, p.L1, p.Lpad1}
#define XR_RAW2(h_) KvRaw<DT, D>{p.k2 + ((int64_t)b * p.L2 * C + (h_) * D) * 2, C, p.vt2 + ((int64_t)(b * H + (h_)) * D * p.Lpad2) * 2, p.L2, p.Lpad2}
#define XR_PK1(h_) KvPacked<DT, D>{p.k1 + (int64_t)(b * H + (h_)) * hb1, p.L1, (p.L1 + 31) & ~31}
#define XR_PK2(h_) KvPacked<DT, D>{p.k2 + (int64_t)(b * H + (h_)) * hb2, p.L2, (p.L2 + 31) & ~31}
#define XR_FETCH1(h_) do { if (pk1) short_load<DT, D, NS1>(f1, XR_PK1(h_), l31, half); else short_load<DT, D, NS1>(f1, XR_RAW1(h_), l31, half); } while (0)
#define XR_FETCH2(h_) do { if (pk2) short_load<DT, D, NSB>(f2, XR_PK2(h_), l31, half); else short_load<DT, D, NSB>(f2, XR_RAW2(h_), l31, half); } while (0)
    // (macros, not lambdas: a fragment struct captured by a lambda is kept in scratch by this compiler)
    if (!(XR_ABL & 2) && !SPLITF) {
        XR_FETCH1((wave & 3) * 2);
        if (DUAL && !BIG2) XR_FETCH2((wave & 3) * 2);
    }
    __syncthreads();
    const float* const bias1 = p.bias1 ? p.bias1 + (int64_t)b * p.L1 : nullptr;
#pragma unroll
    for (int hh = 0; hh < ((XR_ABL & 2) ? 0 : 2); ++hh) {
        const int h = (wave & 3) * 2 + hh;
        if (hh == 1 && !SPLITF) {
            XR_FETCH1(h);
            if (DUAL && !BIG2) XR_FETCH2(h);
        }
#pragma unroll
        for (int pp = 0; pp < PW; ++pp) {
            const int mt = PW == 1 ? (wave >> 2) : pp;
            typename E::v8 qf[KC];
            const uint8_t* qp = Q + (mt * 32 + l31) * ROWB + (h * D + half * 8) * 2;
#pragma unroll
            for (int cc = 0; cc < KC; ++cc) qf[cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(qp + cc * 32));
            f32x16 o[DTT];
#pragma unroll
            for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
            float inv = 1.f;
            if constexpr (SPLITF) {
                if (pk1) short_segment_ns<DT, D, 2>(XR_PK1(h), bias1, p.scale_log2, qf, o, inv, l31, half);
                else short_segment_ns<DT, D, 2>(XR_RAW1(h), bias1, p.scale_log2, qf, o, inv, l31, half);
            } else
                short_compute<DT, D, NS1>(f1, p.L1, bias1, p.scale_log2, qf, o, inv, half);
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
                    if (pk2) long_segment<DT, D>(XR_PK2(h), p.scale_log2, qf, o2, inv2, l31, half);
                    else long_segment<DT, D>(XR_RAW2(h), p.scale_log2, qf, o2, inv2, l31, half);
                } else if constexpr (BIG2 || SPLITF) {
                    constexpr int NSX = BIG2 ? NS2 : 2;
                    if (pk2) short_segment_ns<DT, D, NSX>(XR_PK2(h), nullptr, p.scale_log2, qf, o2, inv2, l31, half);
                    else short_segment_ns<DT, D, NSX>(XR_RAW2(h), nullptr, p.scale_log2, qf, o2, inv2, l31, half);
                } else
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
#pragma unroll
            for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dcol = dt * 32 + 8 * g + 4 * half;
                    if (dcol < D) {
                        typename E::v4 pk;
#pragma unroll
                        for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)o[dt][g * 4 + j];
                        *reinterpret_cast<uint2*>(X + (mt * 32 + l31) * ROWB + (h * D + dcol) * 2) = __builtin_bit_cast(uint2, pk);
                    }
                }
        }
    }
#undef XR_FETCH1
#undef XR_FETCH2
#undef XR_RAW1
#undef XR_RAW2
#undef XR_PK1
#undef XR_PK2
    __syncthreads();

    // ---- 4. to_out(O) + bias -> Q, then + residual -> out ----
    if (!(XR_ABL & 4)) xr_project<DT, C, PW, NSET>(p.wo, X, Q, p.bo, wave, lane);
    __syncthreads();
    uint8_t* const ob = p.out + ((int64_t)b * p.N + row0) * C * 2;
    constexpr int CPR = C / 8;
    for (int idx = tid; idx < XR_TM * CPR; idx += NTH) {
        const int row = idx / CPR, ch = idx - row * CPR;
        if (row >= nrows) break;
        float y[8], r[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(Q + row * ROWB + ch * 16), y);
        unpack8<DT>(*reinterpret_cast<const uint4*>(xb + ((int64_t)row * C + ch * 8) * 2), r);
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] += r[e];
        *reinterpret_cast<uint4*>(ob + ((int64_t)row * C + ch * 8) * 2) = pack8<DT>(y);
    }
}

template <int DT, int C, int NW, int NSET> int xattn_rows_launch(const XrP& p, hipStream_t s) {
    constexpr int LDS = 2 * (NW * 8) * (C * 2 + 16);
    dim3 grid((unsigned)(p.B * p.tiles_per_sample));
    auto go = [&](auto kern) {
        static unsigned devs = 0;
        if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), LDS, &devs) != 0) return -1;
        hipLaunchKernelGGL(kern, grid, dim3(NW * 64), LDS, s, p);
        return apad_check_launch("apad_cross_attention_rows");
    };
    // sub-tile counts of the two segments are compile-time (the fragment registers of an unused sub-tile would not fit beside the rest)
    const int ns1 = p.L1 > 32 ? 2 : 1, ns2 = (p.L2 + 31) / 32;
    // (ns1 == 1: checked by the caller) 8 text + 65 .. 512 audio keys, in 64-key chunks (round 6; the one-tile form of 65 .. 128 keys -- NS2 = 4, its fragments
    //  requested as they are used -- was slower at 128 keys than the chunked form at 256: 50 vs 54 us)
    if (ns2 > 2) return go(xattn_rows_kernel<DT, C, 1, 16, NW, NSET>);
    if (ns1 == 1 && ns2 == 0) return go(xattn_rows_kernel<DT, C, 1, 0, NW, NSET>);
    if (ns1 == 1 && ns2 == 1) return go(xattn_rows_kernel<DT, C, 1, 1, NW, NSET>);
    if (ns2 == 0) return go(xattn_rows_kernel<DT, C, 2, 0, NW, NSET>);
    return go(xattn_rows_kernel<DT, C, 2, 2, NW, NSET>);
}

// apad_rows_pack_kv: one wave per MFMA operand fragment of KvPacked's layout (short_seg.h); blockIdx.y = (sample, head)
__global__ __launch_bounds__(64) void rows_pack_kv_kernel(const uint8_t* k, const uint8_t* vt, uint8_t* out, int H, int D, int L, int Lpad_in) {
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int KC = D / 16, DTT = (D + 31) / 32, NU = (L + 31) / 32, C = H * D;
    const int bh = blockIdx.y, b = bh / H, h = bh - b * H, f = blockIdx.x;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (f < NU * KC) {
        const int u = f / KC, cc = f - u * KC;
        const int key = u * 32 + l31;
        v = *reinterpret_cast<const uint4*>(k + (((int64_t)b * L + (key < L ? key : L - 1)) * C + h * D + cc * 16 + half * 8) * 2);
    } else {
        const int g = f - NU * KC, st = g / DTT, dt = g - st * DTT;
        const int d = dt * 32 + l31;
        if (d < D) {  // (st * 16 + 4 half + 11 < 32 NU <= Lpad_in)
            const uint8_t* vp = vt + (((int64_t)bh * D + d) * Lpad_in + st * 16 + 4 * half) * 2;
            const uint2 v0 = *reinterpret_cast<const uint2*>(vp), v1 = *reinterpret_cast<const uint2*>(vp + 16);
            v = make_uint4(v0.x, v0.y, v1.x, v1.y);
        }
    }
    *reinterpret_cast<uint4*>(out + (int64_t)bh * kv_packed_head_bytes(D, L) + ((int64_t)f * 64 + lane) * 16) = v;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int64_t apad_rows_packed_kv_bytes(int32_t B, int32_t heads, int32_t head_dim, int32_t L) {
    if (B <= 0 || heads <= 0 || head_dim <= 0 || head_dim % 16 != 0 || L <= 0) return 0;
    return (int64_t)B * heads * kv_packed_head_bytes(head_dim, L);
}

extern "C" int apad_rows_pack_kv(const void* k, const void* vt, void* out, int32_t B, int32_t heads, int32_t head_dim, int32_t L, int32_t Lpad, int32_t dtype,
                                 void* stream) {
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_rows_pack_kv: dtype %d not supported (16-bit only)", dtype);
    APAD_CHECK(k && vt && out && B > 0 && heads > 0 && L > 0, "apad_rows_pack_kv: null operand / empty problem");
    APAD_CHECK(head_dim > 0 && head_dim % 16 == 0, "apad_rows_pack_kv: head_dim %d must be a multiple of 16", head_dim);
    APAD_CHECK(Lpad >= L && Lpad % 32 == 0, "apad_rows_pack_kv: Lpad must be >= L and a multiple of 32");
    APAD_CHECK(al16(k) && al16(vt) && al16(out), "apad_rows_pack_kv: pointers must be 16-byte aligned");
    const int nfrag = ((L + 31) / 32) * (head_dim / 16 + 2 * ((head_dim + 31) / 32));
    hipLaunchKernelGGL(rows_pack_kv_kernel, dim3((unsigned)nfrag, (unsigned)(B * heads)), dim3(64), 0, (hipStream_t)stream, (const uint8_t*)k, (const uint8_t*)vt,
                       (uint8_t*)out, heads, head_dim, L, Lpad);
    return apad_check_launch("apad_rows_pack_kv");
}

extern "C" int apad_attention(const apad_attn_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_attention: null descriptor");
    if (d->dtype == APAD_F32) return apad_f32_attention(d, (hipStream_t)stream);  // fp32 precision mode (f32_ops.hip)
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_attention: dtype %d not supported", d->dtype);
    APAD_CHECK(d->q && d->k && d->vt && d->out, "apad_attention: null operand");
    APAD_CHECK(d->B > 0 && d->N > 0 && d->H > 0 && d->L > 0, "apad_attention: empty problem B=%d N=%d H=%d L=%d", d->B, d->N,
               d->H, d->L);
    APAD_CHECK(d->Lpad >= d->L && d->Lpad % 32 == 0, "apad_attention: Lpad must be >= L and a multiple of 32");
    APAD_CHECK(d->kv_batch_div >= 1, "apad_attention: kv_batch_div must be >= 1");
    APAD_CHECK(al16(d->q) && al16(d->k) && al16(d->vt) && al16(d->out) && al16(d->k2) && al16(d->vt2),
               "apad_attention: pointers must be 16-byte aligned");
    APAD_CHECK(d->q_stride_n % 8 == 0 && d->q_stride_b % 8 == 0 && d->k_stride_l % 8 == 0 && d->k_stride_b % 8 == 0 &&
                   d->o_stride_n % 8 == 0 && d->o_stride_b % 8 == 0 && d->vt_stride_b % 8 == 0,
               "apad_attention: strides must keep 16-byte alignment");
    const bool dual = d->L2 > 0;
    if (dual) {
        APAD_CHECK(d->k2 && d->vt2, "apad_attention: segment 2 needs k2/vt2");
        APAD_CHECK(d->Lpad2 >= d->L2 && d->Lpad2 % 32 == 0, "apad_attention: Lpad2 must be >= L2 and a multiple of 32");
        APAD_CHECK(d->kv2_batch_div >= 1, "apad_attention: kv2_batch_div must be >= 1");
        APAD_CHECK(d->k2_stride_l % 8 == 0 && d->k2_stride_b % 8 == 0 && d->vt2_stride_b % 8 == 0,
                   "apad_attention: segment-2 strides must keep 16-byte alignment");
    }
    AttnP p;
    p.q = (const uint8_t*)d->q; p.k = (const uint8_t*)d->k; p.vt = (const uint8_t*)d->vt;
    p.k2 = (const uint8_t*)d->k2; p.vt2 = (const uint8_t*)d->vt2; p.out = (uint8_t*)d->out;
    p.key_bias = d->key_bias;
    p.lse = (float*)d->lse;
    APAD_CHECK(!(dual && d->lse), "apad_attention: lse is only defined for a single softmax segment");
    p.q_sb = d->q_stride_b; p.q_sn = d->q_stride_n; p.k_sb = d->k_stride_b; p.k_sl = d->k_stride_l; p.vt_sb = d->vt_stride_b;
    p.k2_sb = d->k2_stride_b; p.k2_sl = d->k2_stride_l; p.vt2_sb = d->vt2_stride_b; p.o_sb = d->o_stride_b; p.o_sn = d->o_stride_n;
    p.B = d->B; p.N = d->N; p.H = d->H; p.L = d->L; p.Lpad = d->Lpad; p.L2 = d->L2; p.Lpad2 = d->Lpad2;
    p.kvdiv = d->kv_batch_div; p.kvdiv2 = dual ? d->kv2_batch_div : 1;
    p.prescaled = d->q_prescaled ? 1 : 0;
    p.scale_log2 = p.prescaled ? 1.0f : d->softmax_scale * 1.4426950408889634f;
    p.scale2 = d->scale2;
    dim3 grid((unsigned)(((d->N + 127) / 128) * (((d->H * d->B) + 7) / 8 * 8)));
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? launch_dt<APAD_BF16>(p, d->D, dual, grid, s) : launch_dt<APAD_F16>(p, d->D, dual, grid, s);
}

extern "C" int apad_self_attention_fused(const void* x, const void* w_packed, const float* colsum_bias, void* out, int32_t B, int32_t N, int32_t C, int32_t heads,
                                         float ln_eps, int32_t dtype, void* stream) {
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_self_attention_fused: dtype %d not supported (16-bit only)", dtype);
    const bool g256 = C == 256 && heads == 8 && N >= 1 && N <= 1024, g384 = C == 384 && heads == 8 && N >= 1 && N <= 256;
    if (!g256 && !g384) {
        apad_set_error("apad_self_attention_fused: C=%d heads=%d N=%d outside the kernel envelope (256 / 8 / <= 1024, 384 / 8 / <= 256)", C, heads, N);
        return -3;
    }
    APAD_CHECK(x && w_packed && colsum_bias && out && B > 0, "apad_self_attention_fused: null operand / empty batch");
    APAD_CHECK(al16(x) && al16(w_packed) && al16(out) && al16(colsum_bias), "apad_self_attention_fused: pointers must be 16-byte aligned");
    SfP p;
    p.x = (const uint8_t*)x; p.w = (const uint8_t*)w_packed; p.csbb = colsum_bias; p.out = (uint8_t*)out; p.B = B; p.N = N; p.H = heads; p.eps = ln_eps;
    hipStream_t s = (hipStream_t)stream;
    constexpr int NS32 = SF_NSET ? SF_NSET : 4, NS48 = SF_NSET ? SF_NSET : SF_NSET48;
    if (g256) return dtype == APAD_BF16 ? sf_go<APAD_BF16, 32, 16, 8, 2, NS32>(p, s) : sf_go<APAD_F16, 32, 16, 8, 2, NS32>(p, s);
    return dtype == APAD_BF16 ? sf_go<APAD_BF16, 48, 24, 4, SF_NPP48, NS48>(p, s) : sf_go<APAD_F16, 48, 24, 4, SF_NPP48, NS48>(p, s);
}

extern "C" int apad_cross_attention_rows(const apad_xrows_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_cross_attention_rows: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_cross_attention_rows: dtype %d not supported (16-bit only)", d->dtype);
    if (d->C != 384 || d->heads != 8) {  // (the 64-token level, C = 640: apad_hs_attention + apad_hs_out, hsattn.hip)
        apad_set_error("apad_cross_attention_rows: C=%d heads=%d outside the kernel envelope (384, 8)", d->C, d->heads);
        return -3;
    }
    APAD_CHECK(d->x && d->wq_packed && d->wo_packed && d->k1 && d->out, "apad_cross_attention_rows: null operand");
    APAD_CHECK((d->ln_gamma == nullptr) == (d->ln_beta == nullptr), "apad_cross_attention_rows: LayerNorm needs gamma and beta");
    APAD_CHECK(d->B > 0 && d->N > 0, "apad_cross_attention_rows: empty problem B=%d N=%d", d->B, d->N);
    APAD_CHECK(d->L1 >= 1 && d->L1 <= 64 && d->L2 >= 0 && (d->L2 <= 64 || (d->L2 <= 512 && d->L1 <= 32)),
               "apad_cross_attention_rows: segment lengths %d / %d outside 1..64 / 0..64 (0..512 beside <= 32 keys in segment 1)", d->L1, d->L2);
    // (vtN == NULL: kN is the segment's apad_rows_pack_kv set, LpadN unused)
    APAD_CHECK(d->vt1 == nullptr || (d->Lpad1 >= d->L1 && d->Lpad1 % 32 == 0), "apad_cross_attention_rows: Lpad1 must be >= L1 and a multiple of 32");
    const bool dual = d->L2 > 0;
    if (dual) {
        APAD_CHECK(d->k2, "apad_cross_attention_rows: segment 2 needs k2 / vt2 (or its packed set in k2)");
        APAD_CHECK(d->vt2 == nullptr || (d->Lpad2 >= d->L2 && d->Lpad2 % 32 == 0), "apad_cross_attention_rows: Lpad2 must be >= L2 and a multiple of 32");
    }
    APAD_CHECK(al16(d->x) && al16(d->out) && al16(d->wq_packed) && al16(d->wo_packed) && al16(d->k1) && al16(d->vt1) && al16(d->k2) && al16(d->vt2) &&
                   al16(d->ln_gamma) && al16(d->ln_beta),
               "apad_cross_attention_rows: pointers must be 16-byte aligned");
    XrP p;
    p.x = (const uint8_t*)d->x; p.gamma = (const uint8_t*)d->ln_gamma; p.beta = (const uint8_t*)d->ln_beta;
    p.wq = (const uint8_t*)d->wq_packed; p.wo = (const uint8_t*)d->wo_packed; p.bo = (const uint8_t*)d->bo;
    p.k1 = (const uint8_t*)d->k1; p.vt1 = (const uint8_t*)d->vt1; p.bias1 = d->key_bias;
    p.k2 = (const uint8_t*)d->k2; p.vt2 = (const uint8_t*)d->vt2; p.out = (uint8_t*)d->out;
    p.B = d->B; p.N = d->N; p.L1 = d->L1; p.Lpad1 = d->Lpad1; p.L2 = d->L2; p.Lpad2 = d->Lpad2;
    const int tm = 64;  // tokens per workgroup (xattn_rows_kernel, 8 waves)
    p.tiles_per_sample = (d->N + tm - 1) / tm;
    p.eps = d->ln_eps; p.scale_log2 = d->softmax_scale * LOG2E; p.scale2 = d->scale2;
    hipStream_t s = (hipStream_t)stream;
    // (8 waves: 35.8 us at the bench geometry vs 42.5 with 4 waves x 2 panels; deeper weight prefetch -- 6 / 8 register sets -- within 1 us; round 6: 4 waves x
    //  ONE 32-token panel -- 504 workgroups, three resident per CU -- is step-neutral: 34.54 / 34.52 vs 34.58 / 34.55 ms on the same box)
    return d->dtype == APAD_BF16 ? xattn_rows_launch<APAD_BF16, 384, 8, 3>(p, s) : xattn_rows_launch<APAD_F16, 384, 8, 3>(p, s);
}
