// apad_rowpanel_gemm: out[:, seg] = epilogue(LN?(x) . W^T + bias) (+ residual), x panel resident in registers.
//
// Why a second GEMM family: the transformer-block projections have a tiny reduction dim (K = C = 256/384/640) and a
// huge M (64 samples x 1000 tokens).  A 128x128-tiled GEMM re-reads the x tile once per N-tile and restarts its
// pipeline every 4-10 K-steps; here each wave loads its 32 rows of x ONCE (KC x 16 B per lane, as MFMA B-operand
// fragments), optionally LayerNorm-s them in registers (row statistics = in-lane sums + one cross-half exchange), and
// then streams ALL weight rows of the fused projection (q|k|v, or the 8C-wide GEGLU value|gate rows) through a
// double-buffered LDS tile.  x is read from HBM once, LayerNorm costs no pass, the GEGLU product is formed in
// registers (value and gate rows share an MFMA tile: acc[r] pairs with acc[r+8]).
//
// MFMA orientation: D^T[n][m] = sum_k W[n][k] x[m][k]  (A operand = weight rows from LDS, B operand = x fragments),
// so a lane ends up with 4 consecutive output columns of ONE row -> 8-byte stores / residual loads, no LDS epilogue.
// V^T segments use the opposite orientation (lane = channel, 4 consecutive tokens) so the per-head transposed store
// is 8-byte contiguous as well.
//
// LDS: weight tile rows are padded by 16 B (row stride = K*2+16 bytes, an odd number of 16-byte slots), which makes
// the 32-row ds_read_b128 fragment reads bank-conflict free without an XOR swizzle.
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
    static constexpr int BNT = 32;                     // weight rows per LDS tile (one MFMA tile)
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
            const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (m0 + j >= M) break;
                int64_t bj = b;
                int lj = l + j;
                if (lj >= L) {
                    lj -= L;
                    ++bj;
                }
                const uint16_t bits = (uint16_t)(w[j >> 1] >> ((j & 1) * 16));
                reinterpret_cast<uint16_t*>(o)[((bj * heads + h) * hd + dd) * (int64_t)Lpad + lj] = bits;
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

template <int KC, bool GEGLU>
__device__ __forceinline__ void stage_load(u32x4 (&st)[Cfg<KC>::NCH], const uint8_t* w, int64_t ldw, int n_total, int tile, int tid) {
    using C = Cfg<KC>;
#pragma unroll
    for (int i = 0; i < C::NCH; ++i) {
        const int idx = tid + 256 * i;
        const int j = idx / C::CPR, ch = idx - j * C::CPR;
        int64_t row;
        if (GEGLU) {  // LDS tile rows: per 32-row MFMA tile, 16 value rows then the 16 matching gate rows
            const int sub = j >> 5, jj = j & 31;
            const int64_t base = ((int64_t)tile * C::NT + sub) * 16;
            row = jj < 16 ? base + jj : (int64_t)n_total + base + (jj - 16);
        } else {
            row = (int64_t)tile * C::BNT + j;
        }
        st[i] = *reinterpret_cast<const u32x4*>(w + (row * ldw + ch * 8) * 2);
    }
}

template <int KC>
__device__ __forceinline__ void stage_store(const u32x4 (&st)[Cfg<KC>::NCH], uint8_t* base, int tid) {
    using C = Cfg<KC>;
#pragma unroll
    for (int i = 0; i < C::NCH; ++i) {
        const int idx = tid + 256 * i;
        const int j = idx / C::CPR, ch = idx - j * C::CPR;
        *reinterpret_cast<u32x4*>(base + j * C::ROWB + ch * 16) = st[i];
    }
}

template <int DT, int KC, bool LN, bool GEGLU>
__global__ __launch_bounds__(256, 2) void rpgemm_kernel(RpP p) {
    using E = ET<DT>;
    using C = Cfg<KC>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int mt = blockIdx.x / p.nsplit, sp = blockIdx.x - mt * p.nsplit;
    const int t_begin = sp * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.n_tiles);
    if (t_begin >= t_end) return;
    const int64_t mw0 = (int64_t)mt * 128 + wave * 32;  // first row of this wave
    int64_t mrow = mw0 + l31;
    const bool mvalid = mrow < p.M;
    mrow = mvalid ? mrow : p.M - 1;
    constexpr int COLS_PER_TILE = GEGLU ? C::NT * 16 : C::BNT;
    // (batch, token) of the wave's first row, for V^T segments: one 64-bit division per wave instead of one per store
    int64_t vt_b0 = 0;
    int vt_l0 = 0;
    if (!GEGLU && p.L > 0) {
        vt_b0 = mw0 / p.L;
        vt_l0 = (int)(mw0 - vt_b0 * p.L);
    }

    // ---- weight staging (registers -> LDS); first tile's loads are issued after the x panel is in flight ----
    u32x4 st[C::NCH];
    const uint8_t* const wbase = p.w;
    const int64_t ldw = p.ldw;
    const int n_total = p.n_total;

    // ---- x panel -> registers (B-operand fragments), optional LayerNorm ----
    typename E::v8 xf[KC];
    {
        const uint8_t* xp = p.x + (mrow * p.lda + half * 8) * 2;
#pragma unroll
        for (int c = 0; c < KC; ++c) xf[c] = as_v8<DT>(*reinterpret_cast<const uint4*>(xp + c * 32));
    }
    if (LN) {
        // single statistics pass, shifted by the row's first element (both halves of the row use the same shift)
        const float shift = __shfl((float)xf[0][0], l31, 64);
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
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        const float md = s * (1.0f / (KC * 16));
        const float mean = shift + md;
        const float var = fmaxf(q * (1.0f / (KC * 16)) - md * md, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            typename E::v8 g = as_v8<DT>(*reinterpret_cast<const uint4*>(p.gamma + (c * 16 + half * 8) * 2));
            typename E::v8 b = as_v8<DT>(*reinterpret_cast<const uint4*>(p.beta + (c * 16 + half * 8) * 2));
#pragma unroll
            for (int j = 0; j < 8; ++j)
                xf[c][j] = (typename E::elem)(((float)xf[c][j] * rstd + nmr) * (float)g[j] + (float)b[j]);
            asm volatile("" : "+v"(xf[c]) : : "memory");
        }
    }

    stage_load<KC, GEGLU>(st, wbase, ldw, n_total, t_begin, tid);
    stage_store<KC>(st, smem, tid);
    __syncthreads();
    uint8_t* const scr = smem + 2 * C::TILE_BYTES + wave * SCR_BYTES;  // this wave's output transpose scratch
    int cursor = 0, win_col0 = 0;
    // bias of this workgroup's column range -> LDS once (a global bias load per epilogue group put a full memory
    // latency on the critical path of every tile).  Layout: [value cols | gate cols] for GEGLU, fp32.
    float* const lbias = reinterpret_cast<float*>(smem + 2 * C::TILE_BYTES + 4 * SCR_BYTES);
    const int bias_cols = (t_end - t_begin) * COLS_PER_TILE;
    const int bias_c0 = t_begin * COLS_PER_TILE;
    {
        const int reps = GEGLU ? 2 : 1;
        for (int i = tid; i < bias_cols * reps; i += 256) {
            const int part = i / bias_cols, c = i - part * bias_cols;
            const int n = bias_c0 + c;  // global output column
            float v = 0.f;
            if (GEGLU) {
                if (p.seg[0].bias) v = ld_elem<DT>(p.seg[0].bias, (int64_t)part * p.n_total + n);
            } else {
                const bool b1 = p.nseg > 1 && n >= p.seg[1].n_begin, b2 = p.nseg > 2 && n >= p.seg[2].n_begin;
                const uint8_t* bp = b2 ? p.seg[2].bias : (b1 ? p.seg[1].bias : p.seg[0].bias);
                const int nb = b2 ? p.seg[2].n_begin : (b1 ? p.seg[1].n_begin : 0);
                if (bp) v = ld_elem<DT>(bp, n - nb);
            }
            lbias[i] = v;
        }
        __syncthreads();
    }

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) stage_load<KC, GEGLU>(st, wbase, ldw, n_total, t + 1, tid);
        const uint8_t* wt = smem + buf * C::TILE_BYTES + l31 * C::ROWB + half * 16;

        // segment of this tile (tiles never straddle segments: host checks n_cols % COLS_PER_TILE == 0)
        const int n0 = t * COLS_PER_TILE;
        // (explicit selects: a runtime-indexed struct array would be demoted to scratch)
        const bool s1 = p.nseg > 1 && n0 >= p.seg[1].n_begin, s2 = p.nseg > 2 && n0 >= p.seg[2].n_begin;
        Seg sg;
        sg.out = s2 ? p.seg[2].out : (s1 ? p.seg[1].out : p.seg[0].out);
        sg.bias = s2 ? p.seg[2].bias : (s1 ? p.seg[1].bias : p.seg[0].bias);
        sg.ldo = s2 ? p.seg[2].ldo : (s1 ? p.seg[1].ldo : p.seg[0].ldo);
        sg.n_begin = s2 ? p.seg[2].n_begin : (s1 ? p.seg[1].n_begin : p.seg[0].n_begin);
        sg.mode = s2 ? p.seg[2].mode : (s1 ? p.seg[1].mode : p.seg[0].mode);
        const bool vt = (!GEGLU) && sg.mode == APAD_OUT_VT;

        f32x16 acc[C::NT];
#pragma unroll
        for (int s = 0; s < C::NT; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

        if (!vt)
            rp_mainloop<DT, KC, false>(acc, wt, xf);
        else
            rp_mainloop<DT, KC, true>(acc, wt, xf);

        // ---- epilogue: accumulators -> per-wave LDS scratch (transposes the 8-byte-per-lane fragments) -> full
        //      64-byte row segments, 16 bytes per lane.  Writing the fragments straight to HBM (32 rows x 16 B per
        //      store instruction) made the stores, not the MFMAs, the bottleneck of this kernel (measured 3.7x).
        if (GEGLU) {
            // acc[s][4g+j] = value col, acc[s][8+4g+j] = gate col of output column o = n0 + s*16 + 8g + 4half + j
#pragma unroll
            for (int s = 0; s < C::NT; ++s) {
                if (cursor == 0) win_col0 = n0 + s * 16;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int o = n0 + s * 16 + 8 * g + 4 * half;
                    const float4 bv4 = *reinterpret_cast<const float4*>(lbias + (o - bias_c0));
                    const float4 bg4 = *reinterpret_cast<const float4*>(lbias + bias_cols + (o - bias_c0));
                    const float bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w}, bg[4] = {bg4.x, bg4.y, bg4.z, bg4.w};
                    typename E::v4 y;
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const apad_f32x2 gt = {acc[s][8 + 4 * g + j] + bg[j], acc[s][8 + 4 * g + j + 1] + bg[j + 1]};
                        const apad_f32x2 ge = (RP_EXPERIMENT & 2) ? gt : gelu_erf_2(gt);
                        y[j] = (typename E::elem)((acc[s][4 * g + j] + bv[j]) * ge[0]);
                        y[j + 1] = (typename E::elem)((acc[s][4 * g + j + 1] + bv[j + 1]) * ge[1]);
                    }
                    *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (cursor + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                }
                cursor += 16;
                if (cursor == 32) {
                    scratch_flush<DT>(scr, 32, sg.out, sg.ldo, win_col0, nullptr, 0, mw0, p.M, lane);
                    cursor = 0;
                }
            }
        } else if (!vt) {
#pragma unroll
            for (int s = 0; s < C::NT; ++s) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = n0 + s * 32 + 8 * g + 4 * half - sg.n_begin;  // column inside the segment
                    float f[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = acc[s][4 * g + j];
                    {
                        const float4 b4 = *reinterpret_cast<const float4*>(lbias + (n0 + s * 32 + 8 * g + 4 * half - bias_c0));
                        f[0] += b4.x; f[1] += b4.y; f[2] += b4.z; f[3] += b4.w;
                    }
                    if (p.epi == APAD_EPI_SILU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = silu_f(f[j]);
                    } else if (p.epi == APAD_EPI_GELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) f[j] = gelu_erf_f(f[j]);
                    }
                    typename E::v4 y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = (typename E::elem)f[j];
                    *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                }
                scratch_flush<DT>(scr, 32, sg.out, sg.ldo, n0 + s * 32 - sg.n_begin, p.res, p.ldr, mw0, p.M, lane);
            }
        } else {
            // D[m][n]: lane = channel n, registers 4g..4g+3 = 4 consecutive tokens -> scratch[channel][token] ->
            // 64-byte runs of consecutive tokens of one (head, dd) row of V^T
#pragma unroll
            for (int s = 0; s < C::NT; ++s) {
                const float bvv = lbias[n0 + s * 32 + l31 - bias_c0];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typename E::v4 y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = (typename E::elem)(acc[s][4 * g + j] + bvv);
                    *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                }
                scratch_flush_vt<DT>(scr, sg.out, n0 + s * 32 - sg.n_begin, p.heads, p.hd, p.L, p.Lpad, vt_b0, vt_l0, mw0, p.M, lane);
            }
        }

        if (t + 1 < t_end) stage_store<KC>(st, smem + (buf ^ 1) * C::TILE_BYTES, tid);
        __syncthreads();
    }
    if (GEGLU && cursor > 0)  // odd number of 16-column sub-tiles in this workgroup's range
        scratch_flush<DT>(scr, cursor, p.seg[0].out, p.seg[0].ldo, win_col0, nullptr, 0, mw0, p.M, lane);
}

template <int DT, int KC, bool LN, bool GEGLU> int launch(RpP& p, hipStream_t s) {
    using C = Cfg<KC>;
    constexpr int COLS_PER_TILE = GEGLU ? C::NT * 16 : C::BNT;
    p.n_tiles = p.n_total / COLS_PER_TILE;
    const int m_tiles = (int)((p.M + 127) / 128);
    int nsplit = (768 + m_tiles - 1) / m_tiles;  // aim at >= 3 workgroups per CU
    if (nsplit < 1) nsplit = 1;
    if (nsplit > p.n_tiles) nsplit = p.n_tiles;
    p.tiles_per_block = (p.n_tiles + nsplit - 1) / nsplit;
    p.nsplit = (p.n_tiles + p.tiles_per_block - 1) / p.tiles_per_block;
    const size_t lds = 2 * C::TILE_BYTES + 4 * SCR_BYTES + (size_t)p.tiles_per_block * COLS_PER_TILE * (GEGLU ? 2 : 1) * sizeof(float);
    auto kern = rpgemm_kernel<DT, KC, LN, GEGLU>;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_lds = lds;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * p.nsplit)), dim3(256), lds, s, p);
    return apad_check_launch("apad_rowpanel_gemm");
}

template <int DT, int KC> int dispatch2(RpP& p, bool ln, bool geglu, hipStream_t s) {
    if (ln) return geglu ? launch<DT, KC, true, true>(p, s) : launch<DT, KC, true, false>(p, s);
    return geglu ? launch<DT, KC, false, true>(p, s) : launch<DT, KC, false, false>(p, s);
}

template <int DT> int dispatch(RpP& p, int K, bool ln, bool geglu, hipStream_t s) {
    switch (K) {
        case 256: return dispatch2<DT, 16>(p, ln, geglu, s);
        case 384: return dispatch2<DT, 24>(p, ln, geglu, s);
    }
    apad_set_error("apad_rowpanel_gemm: K=%d outside the kernel envelope (256, 384)", K);
    return -3;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline bool al8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

extern "C" int apad_rowpanel_gemm(const apad_rp_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_rowpanel_gemm: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_rowpanel_gemm: dtype %d not supported", d->dtype);
    APAD_CHECK(d->x && d->w && d->M > 0, "apad_rowpanel_gemm: null operand / empty problem");
    APAD_CHECK(d->n_segments >= 1 && d->n_segments <= 3, "apad_rowpanel_gemm: 1..3 segments");
    APAD_CHECK(d->lda % 8 == 0 && d->ldw % 8 == 0 && al16(d->x) && al16(d->w), "apad_rowpanel_gemm: x / w rows must be 16-byte aligned");
    const bool geglu = d->epilogue == APAD_EPI_GEGLU;
    const bool ln = d->ln_gamma != nullptr;
    if (ln) APAD_CHECK(d->ln_beta && al16(d->ln_gamma) && al16(d->ln_beta), "apad_rowpanel_gemm: LayerNorm needs gamma and beta (16-byte aligned)");
    if (d->K != 256 && d->K != 384) {
        apad_set_error("apad_rowpanel_gemm: K=%d outside the kernel envelope (256, 384)", d->K);
        return -3;
    }
    RpP p;
    p.x = (const uint8_t*)d->x; p.w = (const uint8_t*)d->w;
    p.gamma = (const uint8_t*)d->ln_gamma; p.beta = (const uint8_t*)d->ln_beta; p.res = (const uint8_t*)d->residual;
    p.M = d->M; p.lda = d->lda; p.ldw = d->ldw; p.ldr = d->ldr;
    p.epi = d->epilogue; p.nseg = d->n_segments; p.eps = d->ln_eps;
    p.heads = d->heads; p.hd = d->head_dim; p.L = d->L; p.Lpad = d->Lpad;
    int n = 0;
    for (int i = 0; i < 3; ++i) {
        p.seg[i] = Seg{nullptr, nullptr, 0, 0, 0, 0};
        if (i >= d->n_segments) continue;
        const apad_rp_segment& s = d->seg[i];
        APAD_CHECK(s.out && s.n_cols > 0 && s.n_cols % 64 == 0, "apad_rowpanel_gemm: segment %d needs out and n_cols %% 64 == 0", i);
        APAD_CHECK(al16(s.out) && al8(s.bias), "apad_rowpanel_gemm: segment outputs must be 16-byte aligned (bias 8)");
        if (s.mode == APAD_OUT_ROWMAJOR) {
            APAD_CHECK(s.ldo % 8 == 0 && s.ldo >= s.n_cols, "apad_rowpanel_gemm: segment %d ldo must be a multiple of 8 and >= n_cols", i);
        } else if (s.mode == APAD_OUT_VT) {
            APAD_CHECK(!geglu, "apad_rowpanel_gemm: V^T output cannot be combined with GEGLU");
            APAD_CHECK(d->heads > 0 && d->head_dim > 0 && s.n_cols == d->heads * d->head_dim && d->L > 0 && d->Lpad >= d->L &&
                           d->Lpad % 4 == 0 && d->M % d->L == 0,
                       "apad_rowpanel_gemm: V^T segment geometry inconsistent");
        } else {
            apad_set_error("apad_rowpanel_gemm: unknown segment mode %d", s.mode);
            return -1;
        }
        p.seg[i] = Seg{(uint8_t*)s.out, (const uint8_t*)s.bias, s.ldo, n, s.n_cols, s.mode};
        n += s.n_cols;
    }
    p.n_total = n;
    if (geglu) APAD_CHECK(d->n_segments == 1 && d->epilogue == APAD_EPI_GEGLU, "apad_rowpanel_gemm: GEGLU takes one segment");
    if (d->residual)
        APAD_CHECK(d->n_segments == 1 && d->seg[0].mode == APAD_OUT_ROWMAJOR && d->ldr % 8 == 0 && al16(d->residual),
                   "apad_rowpanel_gemm: residual needs a single row-major segment and 8-byte aligned rows");
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? dispatch<APAD_BF16>(p, d->K, ln, geglu, s) : dispatch<APAD_F16>(p, d->K, ln, geglu, s);
}
