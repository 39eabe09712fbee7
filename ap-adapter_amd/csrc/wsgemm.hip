// Weight-stationary row-panel GEMM (the fast path of apad_rowpanel_gemm).
//
// Same contract as rpgemm.hip (LayerNorm? -> x . W^T -> bias / activation / GEGLU / residual, up to 3 column segments,
// V^T output), different schedule: the projection matrices of the transformer blocks are SMALL (K = 256/384, at most
// 2048 rows), the activation is HUGE (64 samples x 1000 tokens).  So a workgroup (8 waves, one per CU) loads a slice of
// NS weight rows into LDS ONCE (132 KB at K=256: 256 rows; 98 KB at K=384: 128 rows) and then every wave streams 32-row
// panels of x through it on its own: x fragments in registers (LayerNorm in registers), KC MFMAs per 32-column tile with
// the A operand read straight from the resident slice, epilogue through the wave's private LDS scratch.  There is NO
// workgroup barrier in the main loop and no per-tile global->LDS staging (in the x-stationary kernel a tile's 16-24
// MFMAs could not cover the latency of the next tile's weight loads); the next panel's x is prefetched into registers
// while the current one is multiplied (K=256).
#include "rp_shared.h"

namespace {

// probe build (tools/ab_build.sh wstr wsgemm.hip -DWS_TRACE=<wave>; tools/ws_trace.py): wall-clock stamps (100 MHz) of one wave of every workgroup.  Never
// part of the product library.
#ifdef WS_TRACE
__device__ unsigned long long ws_trace_buf[1024][16];
#define WS_STAMP(i_) if (lane == 0 && wave == (WS_TRACE) && blockIdx.x < 1024) ws_trace_buf[blockIdx.x][i_] = wall_clock64();
#else
#define WS_STAMP(i_)
#endif

#ifndef APAD_WS_XCD
#define APAD_WS_XCD 1  // (0: the plain block order, for A/B builds)
#endif

template <int KC> struct WsCfg {
    static constexpr int NS = (KC <= 16) ? 256 : 128;  // weight rows resident in LDS per workgroup
    static constexpr int NTILES = NS / 32;             // MFMA tiles per slice
    static constexpr int ROWB = Cfg<KC>::ROWB;         // same padded row stride as the streamed tiles
    static constexpr int CPR = KC * 2;
    static constexpr int W_BYTES = NS * ROWB;
    static constexpr int WAVES = 8;
    static constexpr bool PREFETCH = (KC <= 16);       // second x panel in registers only fits at K=256
};

// RES: a launch with ONE row-major segment and no activation, with or without a residual (to_out / proj_out / proj_in).  These are the launches with about ONE panel per wave (64 x 1000 tokens =
// 2000 panels over 2048 waves): the residual rows of all the wave's tiles are requested with its x panel, behind the weight requests and in front of the
// barrier, instead of one HBM round trip in every tile's epilogue; the registers of the second x panel hold them (the next panel, if any, is loaded at the
// end of the loop).
template <int DT, int KC, bool LN, bool GEGLU, bool RES>
__global__ __launch_bounds__(512) void wsgemm_kernel(RpP p) {
    using E = ET<DT>;
    using W = WsCfg<KC>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int COLS_PER_TILE = GEGLU ? 16 : 32;
    const int nslices = p.nsplit;
    // (row group, weight slice) of this workgroup.  XCD-aware (speed only; round 6): the nslices workgroups of ONE row group share blockIdx % 8, i.e. one
    // XCD's L2 fetches their common x panels once -- with the plain order (slice fastest) the three slices of the 384-wide level sat on three XCDs and
    // each pulled the panels through the fabric: 49.6 MB read per launch for 25 MB of operands (profiles/r06_pmc_traffic_v6.json)
    const int ngrp = gridDim.x / nslices;
    int slice, rgrp;
    {
        const int b = blockIdx.x, full = (ngrp / 8) * 8 * nslices;
        if (APAD_WS_XCD && b < full) {
            const int g = b / (8 * nslices), rem = b - g * 8 * nslices;
            slice = rem >> 3;
            rgrp = g * 8 + (rem & 7);
        } else if (APAD_WS_XCD) {
            const int rem = b - full, tail = ngrp - (ngrp / 8) * 8;
            slice = rem / tail;
            rgrp = (ngrp / 8) * 8 + rem - slice * tail;
        } else {
            slice = b % nslices;
            rgrp = b / nslices;
        }
    }
    const int t0 = slice * W::NTILES;  // first MFMA tile (global tile index) of this slice
    WS_STAMP(0);
    constexpr bool PREFETCH = W::PREFETCH && !RES;
    const int64_t npanels = (p.M + 31) >> 5;
    const int64_t pstride = (int64_t)ngrp * W::WAVES;
    int64_t pi = (int64_t)rgrp * W::WAVES + wave;
    typename E::v8 xf[KC];
    typename E::v8 xn[PREFETCH ? KC : 1];
    const bool has_res = RES && p.res != nullptr;  // (the lean form also serves one-segment launches WITHOUT a residual: proj_in)
    uint4 rall[RES ? W::NTILES : 1][2];  // residual rows of the panel's tiles in scratch_flush_res's lane order (lane, lane + 64 -> (row, 16-byte chunk))
    auto res_load = [&](int64_t mw0_) {
#pragma unroll
        for (int ti_ = 0; ti_ < (RES ? W::NTILES : 0); ++ti_)
#pragma unroll
            for (int k_ = 0; k_ < 2; ++k_) {
                const int idx = lane + 64 * k_, row = idx >> 2, ch = idx & 3;
                int64_t m = mw0_ + row;
                m = m < p.M ? m : p.M - 1;
                rall[ti_][k_] = *reinterpret_cast<const uint4*>(p.res + (m * p.ldr + (t0 + ti_) * 32 - p.seg[0].n_begin + ch * 8) * 2);
            }
    };

    // ---- weight slice + bias -> LDS, once.  Batches of 8 loads are issued back to back before their LDS stores: a plain
    //      load/store loop exposed one full memory latency per 16 bytes per thread (measured: ~20 us of fixed cost). ----
    {
        constexpr int NIT = W::NS * W::CPR / 512;  // chunks per thread (16 at K=256, 12 at K=384)
        static_assert(W::NS * W::CPR % 512 == 0, "slice must split evenly over the workgroup");
        constexpr int UB = 8;
#pragma unroll
        for (int i0 = 0; i0 < NIT; i0 += UB) {
            u32x4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (i0 + u < NIT) {
                    const int idx = tid + 512 * (i0 + u);
                    const int j = idx / W::CPR, ch = idx - j * W::CPR;
                    const int tile = t0 + (j >> 5), jj = j & 31;
                    int64_t row;
                    if (GEGLU) {  // per 32-row MFMA tile: 16 value rows then the 16 matching gate rows
                        const int64_t base = (int64_t)tile * 16;
                        row = jj < 16 ? base + jj : (int64_t)p.n_total + base + (jj - 16);
                    } else {
                        row = (int64_t)tile * 32 + jj;
                    }
                    v[u] = *reinterpret_cast<const u32x4*>(p.w + (row * p.ldw + ch * 8) * 2);
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (i0 + u < NIT) {
                    const int idx = tid + 512 * (i0 + u);
                    const int j = idx / W::CPR, ch = idx - j * W::CPR;
                    *reinterpret_cast<u32x4*>(smem + j * W::ROWB + ch * 16) = v[u];
                }
            }
        }
    }
    if (RES && pi < npanels) {  // behind the weight requests (they are on the critical path), in front of the barrier
        __builtin_amdgcn_sched_barrier(0);
        load_panel<DT, KC>(xf, p.x, p.lda, p.M, pi * 32, l31, half);
        __builtin_amdgcn_sched_barrier(0);
    }
    WS_STAMP(1);
    uint8_t* const scr = smem + W::W_BYTES + wave * SCR_BYTES;
    float* const lbias = reinterpret_cast<float*>(smem + W::W_BYTES + W::WAVES * SCR_BYTES);
    const int bias_cols = W::NTILES * COLS_PER_TILE;
    const int bias_c0 = t0 * COLS_PER_TILE;
    {
        const int reps = GEGLU ? 2 : 1;
        for (int i = tid; i < bias_cols * reps; i += 512) {
            const int part = i / bias_cols, c = i - part * bias_cols;
            const int n = bias_c0 + c;
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
    }
    __syncthreads();  // the only workgroup barrier
    WS_STAMP(2);
    if (has_res && pi < npanels) {  // (behind the barrier: in front of it the 32 requests per lane slowed the weight staging of the whole chip down, 1.4 -> 4.9 us)
        res_load(pi * 32);
        __builtin_amdgcn_sched_barrier(0);
    }

    if (!RES && pi < npanels) load_panel<DT, KC>(xf, p.x, p.lda, p.M, pi * 32, l31, half);

    for (; pi < npanels; pi += pstride) {
        const int64_t mw0 = pi * 32;
        if (LN) layernorm_panel<DT, KC>(xf, p.gamma, p.beta, p.eps, l31, half);
        if constexpr (PREFETCH) {
            if (pi + pstride < npanels) load_panel<DT, KC>(xn, p.x, p.lda, p.M, (pi + pstride) * 32, l31, half);
        }
        int64_t vt_b0 = 0;
        int vt_l0 = 0;
        if (!GEGLU && p.L > 0) {
            vt_b0 = mw0 / p.L;
            vt_l0 = (int)(mw0 - vt_b0 * p.L);
        }
        int cursor = 0, win_col0 = 0;

        if constexpr (RES) {
            // the lean tile loop of the residual launches: one row-major segment, no activation -- none of the general loop's per-tile segment selection,
            // 64-bit row arithmetic and activation branches (the general loop issues ~400 instructions per tile and wave: 1.4-2.0 us per tile, tools/ws_trace.py)
            uint8_t* op[2];
            bool ok[2];
#pragma unroll
            for (int k_ = 0; k_ < 2; ++k_) {
                const int idx = lane + 64 * k_, row = idx >> 2, ch = idx & 3;
                const int64_t m = mw0 + row;
                ok[k_] = m < p.M;
                op[k_] = p.seg[0].out + ((ok[k_] ? m : p.M - 1) * p.seg[0].ldo + (t0 * 32 - p.seg[0].n_begin) + ch * 8) * 2;
            }
            const uint8_t* const sr0 = scr + (lane >> 2) * SCR_ROWB + (lane & 3) * 16;
            static_assert(W::NTILES % 2 == 0, "tiles run in pairs");
#pragma unroll
            for (int tp = 0; tp < W::NTILES; tp += 2) {
                if (tp < 9) { WS_STAMP(3 + tp); }
                // two tiles at a time: 2 independent accumulator chains (a single tile's 16-24 MFMAs are one dependent chain: ~1000 cycles of latency per
                // tile with nothing else to issue); the fragments of a group of 4 k-steps are read together, the next group's reads run under these MFMAs
                f32x16 acc[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
                const uint8_t* const w0p = smem + (tp * 32 + l31) * W::ROWB + half * 16;
#pragma unroll
                for (int g = 0; g < KC / 4; ++g) {
                    typename E::v8 f0[4], f1[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        f0[cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(w0p + (4 * g + cc) * 32));
                        f1[cc] = as_v8<DT>(*reinterpret_cast<const uint4*>(w0p + 32 * W::ROWB + (4 * g + cc) * 32));
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) asm volatile("" : "+v"(f0[cc]), "+v"(f1[cc]) : : "memory");
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        acc[0] = E::mfma32(f0[cc], xf[4 * g + cc], acc[0]);
                        acc[1] = E::mfma32(f1[cc], xf[4 * g + cc], acc[1]);
                    }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int ti = tp + q;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 b4 = *reinterpret_cast<const float4*>(lbias + ti * 32 + 8 * g + 4 * half);
                        typename E::v4 y;
                        y[0] = (typename E::elem)(acc[q][4 * g + 0] + b4.x);
                        y[1] = (typename E::elem)(acc[q][4 * g + 1] + b4.y);
                        y[2] = (typename E::elem)(acc[q][4 * g + 2] + b4.z);
                        y[3] = (typename E::elem)(acc[q][4 * g + 3] + b4.w);
                        *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                    }
#pragma unroll
                    for (int k_ = 0; k_ < 2; ++k_) {
                        const uint2 lo = *reinterpret_cast<const uint2*>(sr0 + k_ * 16 * SCR_ROWB), hi = *reinterpret_cast<const uint2*>(sr0 + k_ * 16 * SCR_ROWB + 8);
                        uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
                        if (has_res) {  // (wave-uniform)
                            float f[8], rr[8];
                            unpack8<DT>(v, f);
                            unpack8<DT>(rall[ti][k_], rr);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += rr[e];
                            v = pack8<DT>(f);
                        }
                        if (ok[k_]) *reinterpret_cast<uint4*>(op[k_] + ti * 64) = v;
                    }
                }
            }
        } else
        for (int ti = 0; ti < W::NTILES; ++ti) {
            if (ti < 9) { WS_STAMP(3 + ti); }
            const int n0 = (t0 + ti) * COLS_PER_TILE;
            const bool s1 = p.nseg > 1 && n0 >= p.seg[1].n_begin, s2 = p.nseg > 2 && n0 >= p.seg[2].n_begin;
            uint8_t* sg_out = s2 ? p.seg[2].out : (s1 ? p.seg[1].out : p.seg[0].out);
            const int64_t sg_ldo = s2 ? p.seg[2].ldo : (s1 ? p.seg[1].ldo : p.seg[0].ldo);
            const int sg_nb = s2 ? p.seg[2].n_begin : (s1 ? p.seg[1].n_begin : p.seg[0].n_begin);
            const int sg_mode = s2 ? p.seg[2].mode : (s1 ? p.seg[1].mode : p.seg[0].mode);
            const bool vt = (!GEGLU) && sg_mode == APAD_OUT_VT;
            const uint8_t* wt = smem + (ti * 32 + l31) * W::ROWB + half * 16;

            f32x16 acc[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
            if (!vt)
                rp_mainloop<DT, KC, false>(acc, wt, xf);
            else
                rp_mainloop<DT, KC, true>(acc, wt, xf);

            if (GEGLU) {
                // acc[4g+j] = value, acc[8+4g+j] = gate of output column o = n0 + 8g + 4half + j
                if (cursor == 0) win_col0 = n0;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int o = n0 + 8 * g + 4 * half;
                    const float4 bv4 = *reinterpret_cast<const float4*>(lbias + (o - bias_c0));
                    const float4 bg4 = *reinterpret_cast<const float4*>(lbias + bias_cols + (o - bias_c0));
                    const float bv[4] = {bv4.x, bv4.y, bv4.z, bv4.w}, bg[4] = {bg4.x, bg4.y, bg4.z, bg4.w};
                    typename E::v4 y;
#pragma unroll
                    for (int j = 0; j < 4; j += 2) {
                        const apad_f32x2 gt = {acc[0][8 + 4 * g + j] + bg[j], acc[0][8 + 4 * g + j + 1] + bg[j + 1]};
                        const apad_f32x2 ge = gelu_erf_2(gt);
                        y[j] = (typename E::elem)((acc[0][4 * g + j] + bv[j]) * ge[0]);
                        y[j + 1] = (typename E::elem)((acc[0][4 * g + j + 1] + bv[j + 1]) * ge[1]);
                    }
                    *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (cursor + 8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                }
                cursor += 16;
                if (cursor == 32) {
                    scratch_flush<DT>(scr, 32, sg_out, sg_ldo, win_col0, nullptr, 0, mw0, p.M, lane);
                    cursor = 0;
                }
            } else if (!vt) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float f[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) f[j] = acc[0][4 * g + j];
                    const float4 b4 = *reinterpret_cast<const float4*>(lbias + (n0 + 8 * g + 4 * half - bias_c0));
                    f[0] += b4.x; f[1] += b4.y; f[2] += b4.z; f[3] += b4.w;
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
                scratch_flush<DT>(scr, 32, sg_out, sg_ldo, n0 - sg_nb, p.res, p.ldr, mw0, p.M, lane);
            } else {
                const float bvv = lbias[n0 + l31 - bias_c0];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typename E::v4 y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = (typename E::elem)(acc[0][4 * g + j] + bvv);
                    *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
                }
                scratch_flush_vt<DT>(scr, sg_out, n0 - sg_nb, p.heads, p.hd, p.L, p.Lpad, vt_b0, vt_l0, mw0, p.M, lane);
            }
        }
        if (GEGLU && cursor > 0) scratch_flush<DT>(scr, cursor, p.seg[0].out, p.seg[0].ldo, win_col0, nullptr, 0, mw0, p.M, lane);
        WS_STAMP(12);

        if constexpr (PREFETCH) {
#pragma unroll
            for (int c = 0; c < KC; ++c) xf[c] = xn[c];
        } else {
            if (pi + pstride < npanels) {
                load_panel<DT, KC>(xf, p.x, p.lda, p.M, (pi + pstride) * 32, l31, half);
                if (has_res) res_load((pi + pstride) * 32);
            }
        }
    }
}

template <int DT, int KC, bool LN, bool GEGLU, bool RES = false> int ws_launch(RpP& p, hipStream_t s) {
    if constexpr (!RES && !GEGLU && !LN) {  // (with the LayerNorm in front the unrolled form does not fit its registers)
        if (p.nseg == 1 && p.seg[0].mode == APAD_OUT_ROWMAJOR && p.epi == APAD_EPI_NONE) return ws_launch<DT, KC, LN, GEGLU, true>(p, s);
    }
    using W = WsCfg<KC>;
    constexpr int COLS_PER_TILE = GEGLU ? 16 : 32;
    const int cols_per_slice = W::NTILES * COLS_PER_TILE;
    if (p.n_total % cols_per_slice != 0) return -3;
    for (int i = 0; i < p.nseg; ++i)
        if (p.seg[i].n_begin % COLS_PER_TILE != 0) return -3;
    p.nsplit = p.n_total / cols_per_slice;  // slices
    const int64_t npanels = (p.M + 31) >> 5;
    int ngrp = 256 / p.nsplit;  // one workgroup per CU
    if (ngrp < 1) ngrp = 1;
    const int64_t max_grp = (npanels + W::WAVES - 1) / W::WAVES;
    if (ngrp > max_grp) ngrp = (int)max_grp;
    const size_t lds = W::W_BYTES + W::WAVES * SCR_BYTES + (size_t)cols_per_slice * (GEGLU ? 2 : 1) * sizeof(float);
    auto kern = wsgemm_kernel<DT, KC, LN, GEGLU, RES>;
    // (the attribute is the per-device ceiling of this instantiation, not the launch's size: set once per device to the CU's whole LDS)
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(ngrp * p.nsplit)), dim3(512), lds, s, p);
    return apad_check_launch("apad_rowpanel_gemm(ws)");
}

template <int DT, int KC> int ws_dispatch2(RpP& p, bool ln, bool geglu, hipStream_t s) {
    if (ln) return geglu ? ws_launch<DT, KC, true, true>(p, s) : ws_launch<DT, KC, true, false>(p, s);
    return geglu ? ws_launch<DT, KC, false, true>(p, s) : ws_launch<DT, KC, false, false>(p, s);
}

}  // namespace

#ifdef WS_TRACE
extern "C" int apad_ws_trace_read(void* dst, int bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(ws_trace_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

// returns -3 when the shape does not fit the weight-stationary schedule (caller falls back to the streamed kernel)
int apad_ws_dispatch(void* rp_params, int K, int dtype, bool ln, bool geglu, void* stream) {
    RpP& p = *reinterpret_cast<RpP*>(rp_params);
    hipStream_t s = (hipStream_t)stream;
    if (K == 256) return dtype == APAD_BF16 ? ws_dispatch2<APAD_BF16, 16>(p, ln, geglu, s) : ws_dispatch2<APAD_F16, 16>(p, ln, geglu, s);
    if (K == 384) return dtype == APAD_BF16 ? ws_dispatch2<APAD_BF16, 24>(p, ln, geglu, s) : ws_dispatch2<APAD_F16, 24>(p, ln, geglu, s);
    return -3;
}
