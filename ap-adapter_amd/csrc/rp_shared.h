// Pieces shared by the two row-panel GEMM kernels (rpgemm.hip: x-stationary with streamed weight tiles;
// wsgemm.hip: weight-stationary persistent): descriptor mirror, per-wave output transpose scratch, the hand-pipelined
// MFMA inner loop.
#pragma once
#include "common.h"

#ifndef RP_EXPERIMENT
#define RP_EXPERIMENT 0  // A/B builds only: 1 = skip global stores, 2 = skip GELU, 3 = both, 4 = skip MFMAs
#endif

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Seg {
    uint8_t* out;
    const uint8_t* bias;
    int64_t ldo;
    int32_t n_begin, n_cols, mode;
};

struct RpP {
    const uint8_t* x;
    const uint8_t* w;
    const uint8_t* gamma;
    const uint8_t* beta;
    const uint8_t* res;
    int64_t M, lda, ldw, ldr;
    int32_t epi, nseg, n_total, n_tiles, tiles_per_block, nsplit;
    float eps;
    int32_t heads, hd, L, Lpad;
    Seg seg[3];
};

template <int KC> struct Cfg {
#ifndef RP_BNT
#define RP_BNT 32
#endif
    static constexpr int BNT = RP_BNT;                 // weight rows per LDS tile (32 = one MFMA tile)
    static constexpr int NT = BNT / 32;               // MFMA tiles per LDS tile
    static constexpr int CPR = KC * 2;                // 16-byte chunks per weight row
    static constexpr int ROWB = KC * 32 + 16;         // padded LDS row stride (bytes)
    static constexpr int TILE_BYTES = BNT * ROWB;
    static constexpr int NCH = BNT * CPR / 256;       // staging chunks per thread
};

constexpr int SCR_ROWB = 72;             // per-wave output scratch: 32 rows x 32 columns, 9 eight-byte slots per row
constexpr int SCR_BYTES = 32 * SCR_ROWB;

// scratch (32 rows x `width` columns, this wave's rows mw0..mw0+31) -> out[:, col0 : col0+width] with 16-byte lanes;
// the optional residual is read with the same coalescing and added after rounding to the storage type (as the
// un-fused reference does).
template <int DT>
__device__ __forceinline__ void scratch_flush(const uint8_t* scr, int width, uint8_t* out, int64_t ldo, int col0,
                                              const uint8_t* res, int64_t ldr, int64_t mw0, int64_t M, int lane) {
    const int cpr = width >> 3;  // 16-byte chunks per row (2 or 4)
    for (int idx = lane; idx < 32 * cpr; idx += 64) {
        const int row = idx / cpr, ch = idx - row * cpr;
        const int64_t m = mw0 + row;
        if (m >= M) continue;
        const uint2 lo = *reinterpret_cast<const uint2*>(scr + row * SCR_ROWB + ch * 16);
        const uint2 hi = *reinterpret_cast<const uint2*>(scr + row * SCR_ROWB + ch * 16 + 8);
        uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (res) {
            float f[8], r[8];
            unpack8<DT>(v, f);
            unpack8<DT>(*reinterpret_cast<const uint4*>(res + (m * ldr + col0 + ch * 8) * 2), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += r[e];
            v = pack8<DT>(f);
        }
        if (!(RP_EXPERIMENT & 1) || v.x == 0x12345678u) *reinterpret_cast<uint4*>(out + (m * ldo + col0 + ch * 8) * 2) = v;
    }
}

// V^T flavour: scratch rows are CHANNELS (nl0 + row), columns are this wave's 32 consecutive tokens (mw0..mw0+31).
// Each 16-byte chunk = 8 consecutive tokens of one (head, dd) row of out[B][heads][hd][Lpad]; chunks that straddle a
// batch boundary or are not 8-aligned inside it fall back to element stores.
template <int DT>
__device__ __forceinline__ void scratch_flush_vt(const uint8_t* scr, uint8_t* out, int nl0, int heads, int hd, int L, int Lpad,
                                                 int64_t b0, int l0, int64_t mw0, int64_t M, int lane) {
    using elem = typename ET<DT>::elem;
    elem* o = reinterpret_cast<elem*>(out);
    for (int idx = lane; idx < 128; idx += 64) {
        const int row = idx >> 2, ch = idx & 3;
        const int nl = nl0 + row;
        const int h = nl / hd, dd = nl - h * hd;
        const int64_t m0 = mw0 + ch * 8;
        if (m0 >= M) continue;
        int64_t b = b0;
        int l = l0 + ch * 8;
        while (l >= L) {
            l -= L;
            ++b;
        }
        const uint2 lo = *reinterpret_cast<const uint2*>(scr + row * SCR_ROWB + ch * 16);
        const uint2 hi = *reinterpret_cast<const uint2*>(scr + row * SCR_ROWB + ch * 16 + 8);
        const int64_t rowoff = ((b * heads + h) * hd + dd) * (int64_t)Lpad;
        if ((l & 7) == 0 && l + 7 < L && m0 + 7 < M) {
            *reinterpret_cast<uint4*>(o + rowoff + l) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            // two 4-token halves (token counts such as 252 are multiples of 4, not 8), element stores as last resort
            const uint2 hv[2] = {lo, hi};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int64_t mq = m0 + 4 * q;
                if (mq >= M) break;
                int64_t bq = b;
                int lq = l + 4 * q;
                if (lq >= L) {
                    lq -= L;
                    ++bq;
                }
                const int64_t ro = ((bq * heads + h) * hd + dd) * (int64_t)Lpad;
                if ((lq & 3) == 0 && lq + 3 < L && mq + 3 < M) {
                    *reinterpret_cast<uint2*>(o + ro + lq) = hv[q];
                } else {
                    const uint32_t w[2] = {hv[q].x, hv[q].y};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (mq + j >= M) break;
                        int64_t bj = bq;
                        int lj = lq + j;
                        if (lj >= L) {
                            lj -= L;
                            ++bj;
                        }
                        reinterpret_cast<uint16_t*>(o)[((bj * heads + h) * hd + dd) * (int64_t)Lpad + lj] =
                            (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
                    }
                }
            }
        }
    }
}

template <int DT, int KC>
__device__ __forceinline__ void rp_load_group(typename ET<DT>::v8 (&wf)[4][Cfg<KC>::NT], const uint8_t* wt, int c0) {
    using C = Cfg<KC>;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int s = 0; s < C::NT; ++s)
            wf[cc][s] = as_v8<DT>(*reinterpret_cast<const uint4*>(wt + s * 32 * C::ROWB + (c0 + cc) * 32));
}

template <int DT, int KC> __device__ __forceinline__ void rp_pin(typename ET<DT>::v8 (&wf)[4][Cfg<KC>::NT]) {
    using C = Cfg<KC>;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
#pragma unroll
        for (int s = 0; s < C::NT; ++s) asm volatile("" : "+v"(wf[cc][s]) : : "memory");
}

// K loop of one weight tile, software-pipelined by hand: the 4*NT weight fragments of chunk group g+1 are requested
// from LDS before the MFMAs of group g issue ("pin" = empty asm that makes the compiler wait for exactly that group's
// reads and forbids sinking the next group's reads below it).  Left alone, hipcc emitted read / wait / MFMA one at a
// time, exposing the LDS latency on every MFMA.  SWAP = false: D^T[n][m] (A = weights, B = x); true: D[m][n].
template <int DT, int KC, bool SWAP>
__device__ __forceinline__ void rp_mainloop(f32x16 (&acc)[Cfg<KC>::NT], const uint8_t* wt, const typename ET<DT>::v8 (&xf)[KC]) {
    using E = ET<DT>;
    using C = Cfg<KC>;
    constexpr int NG = KC / 4;
    typename E::v8 wfa[4][C::NT], wfb[4][C::NT];
    rp_load_group<DT, KC>(wfa, wt, 0);
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
        if (g + 1 < NG) rp_load_group<DT, KC>(wfb, wt, (g + 1) * 4);
        rp_pin<DT, KC>(wfa);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int s = 0; s < C::NT; ++s) {
                if (RP_EXPERIMENT & 4) acc[s][0] += (float)wfa[cc][s][0] * (float)xf[g * 4 + cc][0];
                else acc[s] = SWAP ? E::mfma32(xf[g * 4 + cc], wfa[cc][s], acc[s]) : E::mfma32(wfa[cc][s], xf[g * 4 + cc], acc[s]);
            }
        if (g + 1 < NG) {
            if (g + 2 < NG) rp_load_group<DT, KC>(wfa, wt, (g + 2) * 4);
            rp_pin<DT, KC>(wfb);
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int s = 0; s < C::NT; ++s)
                    acc[s] = SWAP ? E::mfma32(xf[(g + 1) * 4 + cc], wfb[cc][s], acc[s]) : E::mfma32(wfb[cc][s], xf[(g + 1) * 4 + cc], acc[s]);
        }
    }
}

template <int DT, int KC>
__device__ __forceinline__ void load_panel(typename ET<DT>::v8 (&xf)[KC], const uint8_t* x, int64_t lda, int64_t M, int64_t mw0,
                                           int l31, int half) {
    int64_t mrow = mw0 + l31;
    mrow = mrow < M ? mrow : M - 1;
    const uint8_t* xp = x + (mrow * lda + half * 8) * 2;
#pragma unroll
    for (int c = 0; c < KC; ++c) xf[c] = as_v8<DT>(*reinterpret_cast<const uint4*>(xp + c * 32));
}

// RECONVERT: the second pass converts the stored values again instead of keeping the first pass's KC * 8 fp32 values alive (a 128-register budget)
template <int DT, int KC, bool RECONVERT = false>
__device__ __forceinline__ void layernorm_panel(typename ET<DT>::v8 (&xf)[KC], const uint8_t* gamma, const uint8_t* beta, float eps,
                                                int l31, int half) {
    using E = ET<DT>;
    // single statistics pass, shifted by the row's first element (both halves of the row use the same shift)
    const float shift = half_lo((float)xf[0][0]);
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = (float)xf[c][j] - shift;
            s += d;
            q += d * d;
        }
        asm volatile("" : "+v"(s), "+v"(q));  // evaluate chunk by chunk: bounds the live converted values
    }
    s = half_sum(s);
    q = half_sum(q);
    const float md = s * (1.0f / (KC * 16));
    const float mean = shift + md;
    const float var = fmaxf(q * (1.0f / (KC * 16)) - md * md, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float nmr = -mean * rstd;
    if (RECONVERT) {
#pragma unroll
        for (int c = 0; c < KC; ++c) asm volatile("" : "+v"(xf[c]));
    }
#pragma unroll
    for (int c = 0; c < KC; ++c) {
        typename E::v8 g = as_v8<DT>(*reinterpret_cast<const uint4*>(gamma + (c * 16 + half * 8) * 2));
        typename E::v8 b = as_v8<DT>(*reinterpret_cast<const uint4*>(beta + (c * 16 + half * 8) * 2));
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[c][j] = (typename E::elem)(((float)xf[c][j] * rstd + nmr) * (float)g[j] + (float)b[j]);
        asm volatile("" : "+v"(xf[c]) : : "memory");
    }
}

}  // namespace
