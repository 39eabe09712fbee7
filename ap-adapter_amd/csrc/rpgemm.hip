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
#include <stdlib.h>
#include "rp_shared.h"

namespace {

// per-thread byte offsets of the staging loads inside weight tile 0, computed once per kernel; tile t adds a wave-uniform
// stride (the per-tile 64-bit address arithmetic otherwise sits on the vector ALU in front of every tile)
template <int KC, bool GEGLU, int NTH>
__device__ __forceinline__ void stage_offsets(int64_t (&off)[Cfg<KC>::BNT * Cfg<KC>::CPR / NTH], int64_t ldw, int n_total, int tid) {
    using C = Cfg<KC>;
#pragma unroll
    for (int i = 0; i < C::BNT * C::CPR / NTH; ++i) {
        const int idx = tid + NTH * i;
        const int j = idx / C::CPR, ch = idx - j * C::CPR;
        int64_t row;
        if (GEGLU) {  // LDS tile rows: per 32-row MFMA tile, 16 value rows then the 16 matching gate rows
            const int sub = j >> 5, jj = j & 31;
            const int64_t base = (int64_t)sub * 16;
            row = jj < 16 ? base + jj : (int64_t)n_total + base + (jj - 16);
        } else {
            row = j;
        }
        off[i] = (row * ldw + ch * 8) * 2;
    }
}

template <int KC, bool GEGLU, int NTH>
__device__ __forceinline__ void stage_load(u32x4 (&st)[Cfg<KC>::BNT * Cfg<KC>::CPR / NTH], const uint8_t* w, int64_t ldw,
                                           const int64_t (&off)[Cfg<KC>::BNT * Cfg<KC>::CPR / NTH], int tile) {
    using C = Cfg<KC>;
    const uint8_t* base = w + (int64_t)tile * (GEGLU ? C::NT * 16 : C::BNT) * ldw * 2;
#pragma unroll
    for (int i = 0; i < C::BNT * C::CPR / NTH; ++i) st[i] = *reinterpret_cast<const u32x4*>(base + off[i]);
}

template <int KC, int NTH>
__device__ __forceinline__ void stage_store(const u32x4 (&st)[Cfg<KC>::BNT * Cfg<KC>::CPR / NTH], uint8_t* base, int tid) {
    using C = Cfg<KC>;
#pragma unroll
    for (int i = 0; i < C::BNT * C::CPR / NTH; ++i) {
        const int idx = tid + NTH * i;
        const int j = idx / C::CPR, ch = idx - j * C::CPR;
        *reinterpret_cast<u32x4*>(base + j * C::ROWB + ch * 16) = st[i];
    }
}

// NW = waves (32-row panels) per workgroup: 4 (two workgroups per CU) or 8 (one: every weight tile is staged once per CU
// instead of twice)
template <int DT, int KC, bool LN, bool GEGLU, int NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void rpgemm_kernel(RpP p) {
    using E = ET<DT>;
    using C = Cfg<KC>;
    constexpr int NTH = NW * 64, NCHW = C::BNT * C::CPR / NTH;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int mt = blockIdx.x / p.nsplit, sp = blockIdx.x - mt * p.nsplit;
    const int t_begin = sp * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.n_tiles);
    if (t_begin >= t_end) return;
    const int64_t mw0 = (int64_t)mt * (NW * 32) + wave * 32;  // first row of this wave
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
    u32x4 st[NCHW];
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

    int64_t soff[NCHW];
    stage_offsets<KC, GEGLU, NTH>(soff, ldw, n_total, tid);
    stage_load<KC, GEGLU, NTH>(st, wbase, ldw, soff, t_begin);
    stage_store<KC, NTH>(st, smem, tid);
    __syncthreads();
    uint8_t* const scr = smem + 2 * C::TILE_BYTES + wave * SCR_BYTES;  // this wave's output transpose scratch
    int cursor = 0, win_col0 = 0;
    // bias of this workgroup's column range -> LDS once (a global bias load per epilogue group put a full memory
    // latency on the critical path of every tile).  Layout: [value cols | gate cols] for GEGLU, fp32.
    float* const lbias = reinterpret_cast<float*>(smem + 2 * C::TILE_BYTES + NW * SCR_BYTES);
    const int bias_cols = (t_end - t_begin) * COLS_PER_TILE;
    const int bias_c0 = t_begin * COLS_PER_TILE;
    {
        const int reps = GEGLU ? 2 : 1;
        for (int i = tid; i < bias_cols * reps; i += NTH) {
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
        if (t + 1 < t_end) stage_load<KC, GEGLU, NTH>(st, wbase, ldw, soff, t + 1);
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
        if (GEGLU) {
            // b1 is the accumulators' INITIAL value (register 4 g + j: value column, 8 + 4 g + j: its gate), as in the feed-forward kernels
            // (mlp.hip, mlp3.hip) and in geglu3.hip, the form the full-size launches of the 384-wide level take: the same MFMA chain on the same
            // operands -> the same bits, whichever kernel a batch size selects
#pragma unroll
            for (int s = 0; s < C::NT; ++s)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int o = n0 + s * 16 + 8 * g + 4 * half;
                    const float4 bv4 = *reinterpret_cast<const float4*>(lbias + (o - bias_c0));
                    const float4 bg4 = *reinterpret_cast<const float4*>(lbias + bias_cols + (o - bias_c0));
                    acc[s][4 * g + 0] = bv4.x; acc[s][4 * g + 1] = bv4.y; acc[s][4 * g + 2] = bv4.z; acc[s][4 * g + 3] = bv4.w;
                    acc[s][8 + 4 * g + 0] = bg4.x; acc[s][8 + 4 * g + 1] = bg4.y; acc[s][8 + 4 * g + 2] = bg4.z; acc[s][8 + 4 * g + 3] = bg4.w;
                }
        }

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
                    typename E::v4 y;
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const apad_f32x2 gt = {acc[s][8 + 4 * g + j], acc[s][8 + 4 * g + j + 1]};
                        const apad_f32x2 ge = (RP_EXPERIMENT & 2) ? gt : gelu_erf_2(gt);
                        float pr0 = acc[s][4 * g + j] * ge[0], pr1 = acc[s][4 * g + j + 1] * ge[1];  // fp32 product, then ONE rounding (never a v_fma_mix)
                        asm volatile("" : "+v"(pr0), "+v"(pr1));
                        y[j] = (typename E::elem)pr0;
                        y[j + 1] = (typename E::elem)pr1;
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

        if (t + 1 < t_end) stage_store<KC, NTH>(st, smem + (buf ^ 1) * C::TILE_BYTES, tid);
        __syncthreads();
    }
    if (GEGLU && cursor > 0)  // odd number of 16-column sub-tiles in this workgroup's range
        scratch_flush<DT>(scr, cursor, p.seg[0].out, p.seg[0].ldo, win_col0, nullptr, 0, mw0, p.M, lane);
}

template <int DT, int KC, bool LN, bool GEGLU, int NW = 4> int launch(RpP& p, hipStream_t s) {
    using C = Cfg<KC>;
    constexpr int COLS_PER_TILE = GEGLU ? C::NT * 16 : C::BNT;
    if constexpr (NW == 4) {
        constexpr int nw8 = 1;  // step 49.12 -> 48.95 ms
        if (nw8 && p.M >= 256 * 128) return launch<DT, KC, LN, GEGLU, 8>(p, s);
    }
    p.n_tiles = p.n_total / COLS_PER_TILE;
    const int m_tiles = (int)((p.M + NW * 32 - 1) / (NW * 32));
    // Split the column tiles over `nsplit` workgroups per row panel so that the grid is a near-integer number of
    // "rounds" of the workgroups the chip can hold at once (256 CUs x wgs/CU): 1000 workgroups on a 768-slot chip run
    // two rounds for 1.3 rounds of work.  Candidates 1..8, fewest wasted slots wins (ties -> fewer splits).
    const int wg_per_cu = NW == 8 ? 1 : ((GEGLU && KC <= 16) ? 3 : 2);  // from the kernels' VGPR / LDS footprints
    const int cap = 256 * wg_per_cu;
    int nsplit = 1;
    double best = 1e30;
    for (int c = 1; c <= 8 && c <= p.n_tiles; ++c) {
        const int tpb = (p.n_tiles + c - 1) / c;
        const int ns = (p.n_tiles + tpb - 1) / tpb;
        const long blocks = (long)m_tiles * ns;
        const long rounds = (blocks + cap - 1) / cap;
        // time ~ rounds * tiles per block (+ a fixed per-workgroup cost of ~2 tiles for the x panel / LayerNorm)
        const double cost = (double)rounds * (tpb + 2);
        if (cost < best - 1e-9) {
            best = cost;
            nsplit = c;
        }
    }
    p.tiles_per_block = (p.n_tiles + nsplit - 1) / nsplit;
    p.nsplit = (p.n_tiles + p.tiles_per_block - 1) / p.tiles_per_block;
    const size_t lds = 2 * C::TILE_BYTES + NW * SCR_BYTES + (size_t)p.tiles_per_block * COLS_PER_TILE * (GEGLU ? 2 : 1) * sizeof(float);
    auto kern = rpgemm_kernel<DT, KC, LN, GEGLU, NW>;
    // (the attribute is the per-device ceiling of this instantiation, not the launch's size: set once per device to the CU's whole LDS)
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * p.nsplit)), dim3(NW * 64), lds, s, p);
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

int apad_ws_dispatch(void* rp_params, int K, int dtype, bool ln, bool geglu, void* stream);  // wsgemm.hip

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
    // weight-stationary schedule first; APAD_RP_IMPL=stream forces the x-stationary kernel (A/B tests)
    constexpr bool force_stream = false, force_ws = false;
    // measured on MI355X (tools/microbench.py): the weight-stationary schedule wins when all weight rows fit one or a
    // few resident slices (q / to_out / proj_in / proj_out, N = C); fused q|k|v and the 8C-wide GEGLU re-read x once
    // per slice and are faster on the streamed-tile kernel.
    const bool narrow = !geglu && p.n_total <= 512;
    if (!force_stream && (narrow || force_ws)) {
        const int rc = apad_ws_dispatch(&p, d->K, d->dtype, ln, geglu, stream);
        if (rc != -3) return rc;
    }
    return d->dtype == APAD_BF16 ? dispatch<APAD_BF16>(p, d->K, ln, geglu, s) : dispatch<APAD_F16>(p, d->K, ln, geglu, s);
}
