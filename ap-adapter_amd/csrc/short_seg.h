// Short key segments (<= NS x 32 keys in ONE masked tile per segment, no online rescale): the fragment loads and the segment arithmetic
// shared by xattn_rows_kernel (attention.hip) and the head-sliced sub-layer kernels (hsattn.hip).  Textually included INSIDE the
// including file's anonymous namespace, after common.h; needs LOG2E / NEG_BIG defined there.
#pragma once

// Where a segment's MFMA operand fragments come from.  kfrag(u, cc): keys 32 u + lane % 32, features 16 cc + 8 (lane / 32) .. + 8 of the head (keys past L:
// any finite row -- they are masked); vfrag(st, dt): V^T row d = 32 dt + lane % 32, keys 16 st + 4 (lane / 32) + {0..3, 8..11} (zeros for d >= D and past Lpad).
//   KvRaw     apad_attention's layout: k [L][k_sl] row-major (the head's D columns at kbase), vt [D][Lpad] -- every lane of a fragment load sits in its own
//             cache line (32 / 64 lines per instruction)
//   KvPacked  apad_rows_pack_kv's: every fragment one contiguous KB, lane-major (8 lines per instruction; round 6: the fragment loads of a 64-key chunk were
//             ~1000 line look-ups per wave, 5 us per chunk with eight waves on a CU) -- [NU sub-tiles][D / 16] K fragments, then [2 NU][ceil(D / 32)] V^T fragments
template <int DT, int D> struct KvRaw {
    const uint8_t* kbase;
    int64_t k_sl;
    const uint8_t* vbase;
    int L, Lpad;
    __device__ __forceinline__ typename ET<DT>::v8 kfrag(int u, int cc, int l31, int half) const {
        const int key = u * 32 + l31;
        return as_v8<DT>(*reinterpret_cast<const uint4*>(kbase + ((int64_t)(key < L ? key : L - 1) * k_sl + half * 8) * 2 + cc * 32));
    }
    __device__ __forceinline__ typename ET<DT>::v8 vfrag(int st, int dt, int l31, int half) const {
        const int d = dt * 32 + l31;
        uint2 v0 = make_uint2(0u, 0u), v1 = make_uint2(0u, 0u);
        if (d < D && st * 16 < Lpad) {  // (Lpad is a multiple of 32: a visited 16-key step lies inside the padded row)
            const uint8_t* vp = vbase + ((int64_t)d * Lpad + st * 16 + 4 * half) * 2;
            v0 = *reinterpret_cast<const uint2*>(vp);
            v1 = *reinterpret_cast<const uint2*>(vp + 16);
        }
        return as_v8<DT>(make_uint4(v0.x, v0.y, v1.x, v1.y));
    }
};
template <int DT, int D> struct KvPacked {
    const uint8_t* base;  // the (sample, head)'s packed set
    int L, Lpad;          // Lpad = 32 NU
    __device__ __forceinline__ typename ET<DT>::v8 kfrag(int u, int cc, int l31, int half) const {
        const int nu = Lpad >> 5;
        return as_v8<DT>(*reinterpret_cast<const uint4*>(base + (((u < nu ? u : nu - 1) * (D / 16) + cc) * 64 + half * 32 + l31) * 16));
    }
    __device__ __forceinline__ typename ET<DT>::v8 vfrag(int st, int dt, int l31, int half) const {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (st * 16 < Lpad) v = *reinterpret_cast<const uint4*>(base + (Lpad >> 5) * (D / 16) * 1024 + ((st * ((D + 31) / 32) + dt) * 64 + half * 32 + l31) * 16);
        return as_v8<DT>(v);
    }
};
// bytes of one (sample, head)'s packed set
__host__ __device__ constexpr int64_t kv_packed_head_bytes(int D, int L) { return (int64_t)((L + 31) / 32) * (D / 16 + 2 * ((D + 31) / 32)) * 1024; }

// short_segment with the fragment loads split from the arithmetic (xattn_rows_kernel issues the loads of BOTH segments of a head before
// it computes either: one L2 round trip per head instead of four dependent ones).  NS = 32-key sub-tiles of the segment (compile time).
template <int DT, int D, int NS> struct ShortFr {
    typename ET<DT>::v8 kf[NS][D / 16];
    typename ET<DT>::v8 vf[2 * NS][(D + 31) / 32];
};
template <int DT, int D, int NS, class Src>
__device__ __forceinline__ void short_load(ShortFr<DT, D, NS>& f, const Src& src, int l31, int half) {
    constexpr int KC = D / 16, DTT = (D + 31) / 32;
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) f.kf[u][cc] = src.kfrag(u, cc, l31, half);  // rows past L: masked in short_compute
#pragma unroll
    for (int st = 0; st < 2 * NS; ++st)
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt) f.vf[st][dt] = src.vfrag(st, dt, l31, half);
}
template <int DT, int D, int NS>
__device__ __forceinline__ void short_compute(const ShortFr<DT, D, NS>& f, int L, const float* bias, float c, const typename ET<DT>::v8* qf, f32x16* o,
                                              float& inv_den, int half) {
    using E = ET<DT>;
    constexpr int KC = D / 16, DTT = (D + 31) / 32;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        s[u] = zero16;
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) s[u] = E::mfma32(f.kf[u][cc], qf[cc], cc == 0 ? zero16 : s[u]);
    }
    float tmax = NEG_BIG;
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = s[u][r] * c;
            if (bias) v += bias[key < L ? key : L - 1] * LOG2E;
            v = key < L ? v : NEG_BIG;
            s[u][r] = v;
            tmax = fmaxf(tmax, v);
        }
    tmax = half_max(tmax);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = (float)(typename E::elem)__builtin_amdgcn_exp2f(s[u][r] - tmax);  // (as short_segment: the sum of the ROUNDED probabilities)
            s[u][r] = e;
            sum += e;
        }
    sum = half_sum(sum);
    inv_den = 1.0f / sum;
#pragma unroll
    for (int st = 0; st < 2 * NS; ++st) {
        typename E::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[st >> 1][(st & 1) * 8 + j];
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt) o[dt] = E::mfma32(f.vf[st][dt], pf, o[dt]);
    }
}

// the same arithmetic with the fragments requested where they are used (no resident fragment set): segments of up to NS x 32 keys --
// the 128 audio keys of the timbre / accompaniment presets -- whose K and V^T fragments (64 + 64 registers at NS = 4) do not fit beside
// the rest of xattn_rows_kernel's attention phase
template <int DT, int D, int NS, class Src>
__device__ __forceinline__ void short_segment_ns(const Src& src, const float* bias, float c, const typename ET<DT>::v8* qf, f32x16* o, float& inv_den, int l31,
                                                 int half) {
    const int L = src.L;
    using E = ET<DT>;
    constexpr int KC = D / 16, DTT = (D + 31) / 32;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        s[u] = zero16;
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) s[u] = E::mfma32(src.kfrag(u, cc, l31, half), qf[cc], cc == 0 ? zero16 : s[u]);
    }
    float tmax = NEG_BIG;
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = s[u][r] * c;
            if (bias) v += bias[key < L ? key : L - 1] * LOG2E;
            v = key < L ? v : NEG_BIG;
            s[u][r] = v;
            tmax = fmaxf(tmax, v);
        }
    tmax = half_max(tmax);
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < NS; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = (float)(typename E::elem)__builtin_amdgcn_exp2f(s[u][r] - tmax);
            s[u][r] = e;
            sum += e;
        }
    sum = half_sum(sum);
    inv_den = 1.0f / sum;
#pragma unroll
    for (int st = 0; st < 2 * NS; ++st) {
        typename E::v8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[st >> 1][(st & 1) * 8 + j];
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt) o[dt] = E::mfma32(src.vfrag(st, dt, l31, half), pf, o[dt]);
    }
}

// LONG second segments (65 .. 512 audio keys: the timbre / accompaniment presets, pooling 1 and the mixed poolings of the cfg-3 sweep, AudioMAE.py:148-182) in
// the same kernels: the segment runs in 64-key chunks with a running maximum / sum, the accumulator rescaled when the maximum moves (the flash form
// apad_attention and the chunked xattn_kernel use for such lengths: maximum in the raw score domain, the scale folded into the exponent's fma,
// UN-normalised probabilities rounded to the storage type for the P.V product, one division at the end -- the caller's o2 * inv_den).  The K fragments
// of chunk i + 1 are requested behind chunk i's scores, a chunk's V^T fragments in front of them: one exposed L2 round trip per segment, not per chunk.
// Only a ragged last chunk pays for the key mask.  No key bias (the audio segment never has one).
template <int DT, int D, class Src>
__device__ __forceinline__ void long_segment(const Src& src, float c, const typename ET<DT>::v8* qf, f32x16* o, float& inv_den, int l31, int half) {
    using E = ET<DT>;
    constexpr int KC = D / 16, DTT = (D + 31) / 32;
    const int L = src.L;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    typename E::v8 kf[2][KC];  // ONE fragment set, refilled in place as soon as the chunk's score MFMAs have read it
    auto load_k = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int cc = 0; cc < KC; ++cc) kf[u][cc] = src.kfrag((k0 >> 5) + u, cc, l31, half);  // rows past L: masked below
    };
    load_k(0);
    float m = NEG_BIG, l0 = 0.f, l1 = 0.f;
    for (int k0 = 0; k0 < L; k0 += 64) {
        typename E::v8 vf[4][DTT];
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int dt = 0; dt < DTT; ++dt) vf[st][dt] = src.vfrag((k0 >> 4) + st, dt, l31, half);
        f32x16 s[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            s[u] = zero16;
#pragma unroll
            for (int cc = 0; cc < KC; ++cc) s[u] = E::mfma32(kf[u][cc], qf[cc], cc == 0 ? zero16 : s[u]);
        }
        load_k(k0 + 64);  // the next chunk's, under this one's softmax + P.V (past the end: a clamped sub-tile, never used)
        if (k0 + 64 > L) {  // (wave-uniform) the ragged last chunk: keys past L out of the maximum and the sums
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= L) s[u][r] = NEG_BIG;
        }
        float tmax = s[0][0];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[u][r]);
        const float mnew = fmaxf(m, half_max(tmax));                  // (chunk 0 holds key 0: finite from the first chunk on)
        const float alpha = __builtin_amdgcn_exp2f((m - mnew) * c);   // c > 0; the first chunk: exp2(-huge) = 0 against a zero accumulator
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
        l0 = __builtin_fmaf(l0, alpha, sum0);  // per-lane partial sums of the query's row (alpha is the same in both half-waves)
        l1 = __builtin_fmaf(l1, alpha, sum1);
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            typename E::v8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[st >> 1][(st & 1) * 8 + j];
#pragma unroll
            for (int dt = 0; dt < DTT; ++dt) o[dt] = E::mfma32(vf[st][dt], pf, o[dt]);
        }
        m = mnew;
    }
    inv_den = 1.0f / half_sum(l0 + l1);
}
