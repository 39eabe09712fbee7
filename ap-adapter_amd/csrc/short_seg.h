// Short key segments (<= NS x 32 keys in ONE masked tile per segment, no online rescale): the fragment loads and the segment arithmetic
// shared by xattn_rows_kernel (attention.hip) and the head-sliced sub-layer kernels (hsattn.hip).  Textually included INSIDE the
// including file's anonymous namespace, after common.h; needs LOG2E / NEG_BIG defined there.
#pragma once

// short_segment with the fragment loads split from the arithmetic (xattn_rows_kernel issues the loads of BOTH segments of a head before
// it computes either: one L2 round trip per head instead of four dependent ones).  NS = 32-key sub-tiles of the segment (compile time).
template <int DT, int D, int NS> struct ShortFr {
    typename ET<DT>::v8 kf[NS][D / 16];
    typename ET<DT>::v8 vf[2 * NS][(D + 31) / 32];
};
template <int DT, int D, int NS>
__device__ __forceinline__ void short_load(ShortFr<DT, D, NS>& f, const uint8_t* kbase, int64_t k_sl, const uint8_t* vbase, int L, int Lpad, int l31,
                                           int half) {
    constexpr int KC = D / 16, DTT = (D + 31) / 32;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int key = u * 32 + l31;
        const uint8_t* kp = kbase + ((int64_t)(key < L ? key : L - 1) * k_sl + half * 8) * 2;  // rows past L: masked in short_compute
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) f.kf[u][cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(kp + cc * 32));
    }
#pragma unroll
    for (int st = 0; st < 2 * NS; ++st) {
        const int kcol = st * 16 + 4 * half;
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt) {
            const int d = dt * 32 + l31;
            uint2 v0 = make_uint2(0u, 0u), v1 = make_uint2(0u, 0u);
            if (d < D && st * 16 < Lpad) {  // (Lpad is a multiple of 32: a visited 16-key step lies inside the padded row)
                const uint8_t* vp = vbase + ((int64_t)d * Lpad + kcol) * 2;
                v0 = *reinterpret_cast<const uint2*>(vp);
                v1 = *reinterpret_cast<const uint2*>(vp + 16);
            }
            f.vf[st][dt] = as_v8<DT>(make_uint4(v0.x, v0.y, v1.x, v1.y));
        }
    }
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
template <int DT, int D, int NS>
__device__ __forceinline__ void short_segment_ns(const uint8_t* kbase, int64_t k_sl, const uint8_t* vbase, int L, int Lpad, const float* bias, float c,
                                                 const typename ET<DT>::v8* qf, f32x16* o, float& inv_den, int l31, int half) {
    using E = ET<DT>;
    constexpr int KC = D / 16, DTT = (D + 31) / 32;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 s[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const int key = u * 32 + l31;
        const uint8_t* kp = kbase + ((int64_t)(key < L ? key : L - 1) * k_sl + half * 8) * 2;
        s[u] = zero16;
#pragma unroll
        for (int cc = 0; cc < KC; ++cc) s[u] = E::mfma32(as_v8<DT>(*reinterpret_cast<const uint4*>(kp + cc * 32)), qf[cc], cc == 0 ? zero16 : s[u]);
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
        const int kcol = st * 16 + 4 * half;
#pragma unroll
        for (int dt = 0; dt < DTT; ++dt) {
            const int d = dt * 32 + l31;
            uint2 v0 = make_uint2(0u, 0u), v1 = make_uint2(0u, 0u);
            if (d < D && st * 16 < Lpad) {
                const uint8_t* vp = vbase + ((int64_t)d * Lpad + kcol) * 2;
                v0 = *reinterpret_cast<const uint2*>(vp);
                v1 = *reinterpret_cast<const uint2*>(vp + 16);
            }
            o[dt] = E::mfma32(as_v8<DT>(make_uint4(v0.x, v0.y, v1.x, v1.y)), pf, o[dt]);
        }
    }
}
