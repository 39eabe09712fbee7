// apad_gemm: out = epilogue(A . W^T + bias + rowgroup_bias) + residual on MFMA 32x32x16 (bf16/f16, fp32 acc).
//
// One kernel family covers every dense contraction of the path: Linear layers, 1x1 and 3x3
// convolutions as implicit GEMM over NHWC activations (the im2col gather happens while staging the A tile),
// the AudioMAE patch embedding (16x16 fp32 mel patches gathered and converted while staging), GEGLU
// (value|gate weight rows interleaved per block so a*gelu(g) is formed in the epilogue and the 8C-wide
// intermediate never reaches HBM) and the per-head transposed V^T store apad_attention consumes.
//
// Tiling: 128x128 block tile, BK=64, 4 waves (2x2), each wave 64x64 = 2x2 MFMA 32x32x16 tiles.
// Staging: global -> registers (16 B/lane, coalesced along K) -> XOR-swizzled LDS (conflict-free
// ds_read_b128 fragment reads); next K-tile's global loads are issued before the MFMAs of the current one.
// Epilogue: accumulators -> LDS tile -> full-row 16 B stores (residual read with the same coalescing).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "f32_ops.h"

namespace {

constexpr int BK = 64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// TM = block tile edge: 128 (4 waves x 64x64) for big problems, 64 (4 waves x 32x32) when a 128-tiling would leave
// most of the 256 CUs idle (the 64-token / 4096-row level of the UNet, 32x2 convolutions)
// KG = K groups: KG x 4 waves per workgroup, group g reduces the k-tiles g, g + KG, ... of the SAME output tile from its own LDS
// stages and the partial accumulators are summed through LDS before the epilogue (in-workgroup split-K).  For the latency-bound
// launches of the 640- / 384-wide levels (a few hundred 64x64 tiles with 6..40 k-tiles each): the serial k-loop per workgroup is
// what those launches wait for, and a second wave per SIMD covers the first one's LDS round trips.
template <int TM, int NS = 1, int KG = 1> struct Tile {
    static constexpr int BM = TM, BN = TM;
    static constexpr int MI = TM / 64;              // MFMA tiles per wave per dimension
    static constexpr int A_BYTES = TM * BK * 2;
    static constexpr int C_LD = TM + 8;             // epilogue tile row stride (elements)
    static constexpr int STAGE = 2 * A_BYTES;       // one k-tile of A and of W; NS stages
    static constexpr int RED_BYTES = (KG > 1) ? (KG - 1) * 256 * MI * MI * 16 * 4 : 0;  // fp32 partials of groups 1..KG-1
    static constexpr int CT_BYTES = TM * C_LD * 2;
    // epilogue tile at offset 0; the partials behind it (both are used after the last k-loop barrier, when the stages are dead)
    static constexpr int EPI_BYTES = (KG > 1) ? ((CT_BYTES + 15) / 16 * 16 + RED_BYTES) : CT_BYTES;
    static constexpr int SMEM_BYTES = (EPI_BYTES > KG * NS * STAGE) ? EPI_BYTES : KG * NS * STAGE;
    static constexpr int NLD = TM / 32;             // staging vectors per thread per operand
};

struct GemmP {
    const uint8_t* a;
    const uint8_t* w;
    uint8_t* out;
    uint8_t* out2;
    uint8_t* out3;
    uint8_t* out4;  // APAD_OUT_QKV: optional row-major v
    const uint8_t* bias;
    const uint8_t* residual;
    const uint8_t* rg;
    const int32_t* step_ptr;
    int64_t M, N, K, lda, ldw, ldo, ldr, ld_rg, rows_per_group;
    int32_t Hin, Win, Cin, Hout, Wout, stride, Hup, Wup, src_batch_mod, res_mod;
    int32_t heads, head_dim, L, Lpad;
    int32_t wrows;  // rows of w (N, or 2N for GEGLU)
    int32_t m_tiles, n_tiles;
    int32_t taps, dilation, pad, transposed, pre_act;  // APAD_A_CONV1D
    float pre_slope;
    int32_t lead;  // conv3x3: zero rows / columns before the first source row / column (1, or 0 with conv_asym_pad)
    // LayerNorm folded into the contraction (apad_gemm_desc::rowstat_in): a = RAW rows, w = gamma-scaled weights,
    // out = rstd_m * (acc - mean_m * ln_cs[n]) + ln_bb[n]; the row statistics are summed from the producing kernel's partials
    float* rs_out;        // [M][rs_out_tiles][2]: per 64-column block (sum, sum of squares) of the stored output row
    const float* rs_in;   // [M][rs_in_tiles][2]
    const float* ln_cs;   // [w rows]
    const float* ln_bb;   // [w rows]
    int32_t rs_in_tiles, rs_out_tiles;
    float ln_eps;
    // two-source plain A (apad_gemm_desc::a2): columns >= ksplit of row m come from a2[(m % a2_mod) * lda2 + k - ksplit]
    const uint8_t* a2;
    int64_t lda2;
    int32_t ksplit, a_mod, a2_mod;
};

// byte offset of 16-byte chunk `chunk` (0..7) of tile row `row` (128-byte rows)
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// internal A mode: 3x3 implicit conv whose every 64-wide k-tile lies inside ONE filter tap (Cin % 64 == 0) and whose
// source is not upsampled: the tap / channel split of k is tracked by scalar counters instead of per-chunk integer
// divisions, and each row keeps a precomputed element offset (the gather was 25 % of the conv kernel's wave cycles in VALU)
#define APAD_A_CONV3X3_FAST 3

template <int AMODE> struct RowInfo {
    int64_t base;  // PLAIN: element offset of the row; CONV/PATCH: source batch index
    int oy, ox;
    bool valid;
};

template <int DT, int AMODE>
__device__ __forceinline__ uint4 load_a(const GemmP& p, const RowInfo<AMODE>& r, int k) {
    uint4 z = make_uint4(0, 0, 0, 0);
    if (!r.valid || k >= p.K) return z;
    if (AMODE == APAD_A_PLAIN) {
        if (p.a2 != nullptr && k >= p.ksplit) {  // (r.oy = the row index: the second source's offset is formed here)
            const int64_t m2 = p.a2_mod > 0 ? r.oy % p.a2_mod : r.oy;
            return *reinterpret_cast<const uint4*>(p.a2 + (m2 * p.lda2 + (k - p.ksplit)) * 2);
        }
        return *reinterpret_cast<const uint4*>(p.a + (r.base + k) * 2);
    } else if (AMODE == APAD_A_CONV3X3) {
        int tap = k / p.Cin;
        int c = k - tap * p.Cin;
        int ky = tap / 3, kx = tap - ky * 3;
        int iy = r.oy * p.stride + ky - p.lead, ix = r.ox * p.stride + kx - p.lead;
        int H = p.Hup > 0 ? p.Hup : p.Hin, W = p.Hup > 0 ? p.Wup : p.Win;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) return z;
        if (p.Hup > 0) {  // nearest-neighbour source index, floor(dst * in / out)
            iy = (int)(((int64_t)iy * p.Hin) / p.Hup);
            ix = (int)(((int64_t)ix * p.Win) / p.Wup);
        }
        int64_t off = ((r.base * p.Hin + iy) * p.Win + ix) * p.Cin + c;
        return *reinterpret_cast<const uint4*>(p.a + off * 2);
    } else if (AMODE == APAD_A_CONV1D) {  // channels-last [B][Hin][Cin]; r.base = b, r.oy = t; k = (tap, c)
        const int tap = k / p.Cin, c = k - tap * p.Cin;
        int ti;
        if (p.transposed) {
            const int num = r.oy + p.pad - tap;
            ti = num / p.stride;
            if (num < 0 || ti * p.stride != num) return z;
        } else {
            ti = r.oy + tap * p.dilation - p.pad;
        }
        if (ti < 0 || ti >= p.Hin) return z;
        uint4 v = *reinterpret_cast<const uint4*>(p.a + (((int64_t)r.base * p.Hin + ti) * p.Cin + c) * 2);
        if (p.pre_act) {  // the vocoder's pre-activation, applied while staging
            float f[8];
            unpack8<DT>(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * p.pre_slope;
            v = pack8<DT>(f);
        }
        return v;
    } else {  // PATCH16: fp32 mel [B][Hin][Win]; k = py*16 + px
        int py = k >> 4, px = k & 15;
        int64_t off = (r.base * p.Hin + r.oy * 16 + py) * p.Win + r.ox * 16 + px;
        const float4* src = reinterpret_cast<const float4*>(p.a + off * 4);
        float4 f0 = src[0], f1 = src[1];
        float f[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
        return pack8<DT>(f);
    }
}

template <int DT, int AMODE, int EPI, int OUTMODE, int TM, int NS = 1, int KG = 1>
__global__ __launch_bounds__(256 * KG) void gemm_kernel(GemmP p) {
    using T = Tile<TM, NS, KG>;
    constexpr int BM = T::BM, BN = T::BN, MI = T::MI, A_BYTES = T::A_BYTES, C_LD = T::C_LD, NLD = T::NLD, WT = TM / 2;
    constexpr int NT = 256 * KG;
    __shared__ __attribute__((aligned(16))) uint8_t smem[T::SMEM_BYTES];
    using E = ET<DT>;
    // tid = thread index inside its K group (the staging / fragment roles of the 4-wave kernel); kg = K group (wave-uniform)
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int kg = (KG > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
    const int gso = kg * NS * T::STAGE;  // this group's LDS stages
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int BN_OUT = (EPI == APAD_EPI_GEGLU) ? BN / 2 : BN;
    constexpr int GH = BN / 2;  // GEGLU: first half of the tile columns = value rows, second half = gate rows
    // XCD-aware tile order: workgroup id b runs on XCD b % 8 (observed dispatch order, speed only).  Tiles are
    // numbered so that all N-tiles of one M-tile share b % 8, i.e. one XCD's L2 fetches each A row-panel once.
    int mt, nt;
    {
        const int nN = p.n_tiles, nM = p.m_tiles;
        const int b = blockIdx.x;
        const int full = (nM / 8) * 8 * nN;  // blocks covered by complete groups of 8 M-tiles
        if (b < full) {
            const int grp = b / (8 * nN), rem = b - grp * 8 * nN;
            nt = rem >> 3;
            mt = grp * 8 + (rem & 7);
        } else {
            const int rem = b - full, tail = nM - (nM / 8) * 8;  // < 8 leftover M-tiles
            nt = rem / tail;
            mt = (nM / 8) * 8 + rem - nt * tail;
        }
    }
    const int64_t m0 = (int64_t)mt * BM;
    const int64_t n0 = (int64_t)nt * BN_OUT;

    // W row feeding local tile column nl
    auto wrow = [&](int nl) -> int64_t {
        if (EPI == APAD_EPI_GEGLU) return nl < GH ? n0 + nl : p.N + n0 + (nl - GH);
        return n0 + nl;
    };
    auto wrow_valid = [&](int nl) -> bool {
        if (EPI == APAD_EPI_GEGLU) return (nl < GH ? n0 + nl : n0 + nl - GH) < p.N;
        return n0 + nl < p.N;
    };

    // LayerNorm-by-algebra: mean / rstd of this tile's rows, summed in a fixed order from the producer's 64-column partials
    __shared__ float rstat[TM][2];
    if (p.rs_in != nullptr && threadIdx.x < BM) {
        const int64_t m = m0 + threadIdx.x;
        float s1 = 0.f, s2 = 0.f;
        if (m < p.M) {
            const float* src = p.rs_in + m * p.rs_in_tiles * 2;
            for (int t_ = 0; t_ < p.rs_in_tiles; ++t_) {
                s1 += src[2 * t_];
                s2 += src[2 * t_ + 1];
            }
        }
        const float mean = s1 / (float)p.K;
        const float var = fmaxf(s2 / (float)p.K - mean * mean, 0.f);
        rstat[threadIdx.x][0] = mean;
        rstat[threadIdx.x][1] = rsqrtf(var + p.ln_eps);
    }
    // per-thread staging assignment: rows (tid>>3) + 32*i, 16-byte chunk tid&7
    const int chunk = tid & 7;
    RowInfo<AMODE> ra[NLD];
    int64_t wb[NLD];
    bool wv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int rl = (tid >> 3) + 32 * i;
        int64_t m = m0 + rl;
        ra[i].valid = m < p.M;
        ra[i].oy = ra[i].ox = 0;
        ra[i].base = 0;
        if (ra[i].valid) {
            if (AMODE == APAD_A_PLAIN) {
                ra[i].base = (p.a_mod > 0 ? m % p.a_mod : m) * p.lda;
                ra[i].oy = (int)m;
            } else if (AMODE == APAD_A_CONV3X3) {
                int64_t hw = (int64_t)p.Hout * p.Wout;
                int64_t b = m / hw;
                int rem = (int)(m - b * hw);
                ra[i].oy = rem / p.Wout;
                ra[i].ox = rem - ra[i].oy * p.Wout;
                ra[i].base = p.src_batch_mod > 0 ? b % p.src_batch_mod : b;
            } else if (AMODE == APAD_A_CONV1D) {
                const int64_t b = m / p.Hout;
                ra[i].oy = (int)(m - b * p.Hout);
                ra[i].base = b;
            } else if (AMODE == APAD_A_CONV3X3_FAST) {
                int64_t hw = (int64_t)p.Hout * p.Wout;
                int64_t b = m / hw;
                int rem = (int)(m - b * hw);
                const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
                ra[i].oy = oy * p.stride - p.lead;  // source row / column of filter tap (0, 0)
                ra[i].ox = ox * p.stride - p.lead;
                const int64_t sb = p.src_batch_mod > 0 ? b % p.src_batch_mod : b;
                ra[i].base = ((sb * p.Hin + ra[i].oy) * p.Win + ra[i].ox) * p.Cin;  // element offset of that tap, channel 0
            } else {
                int wp = p.Win >> 4, hp = p.Hin >> 4;
                int64_t b = m / (hp * wp);
                int rem = (int)(m - b * hp * wp);
                ra[i].oy = rem / wp;
                ra[i].ox = rem - ra[i].oy * wp;
                ra[i].base = b;
            }
        }
        wv[i] = wrow_valid(rl);
        wb[i] = wv[i] ? wrow(rl) * p.ldw : 0;
    }

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = (int)((p.K + BK - 1) / BK);
    // Global -> register staging, PF k-tiles ahead of the LDS copy.  Depths 2 and 3 were measured on the 64x64 variant
    // (the 640- and 384-wide levels): no gain (10.5 / 11.2 us vs 10.7 us at M=4096, K=640; 41.8 / 44.1 vs 39.9 us at
    // M=16128, K=1536) -- those launches are bound by the two barriers per k-tile, not by load latency -- so PF stays 1.
#ifndef APAD_GEMM_PF64
#define APAD_GEMM_PF64 1
#endif
    constexpr int PF = (TM == 64) ? APAD_GEMM_PF64 : 1;
    u32x4 ga[PF][NLD], gb[PF][NLD];  // (native vectors: HIP's uint4 class type kept such arrays in scratch)
    int ftap = 0, fc0 = 0;  // CONV3X3_FAST: filter tap and first channel of the NEXT k-tile to load (tiles load in order)
    auto gload = [&](int kt, u32x4* gA, u32x4* gB) {
        int k = kt * BK + chunk * 8;
        int fky = 0, fkx = 0;
        int64_t fkoff = 0;
        if (AMODE == APAD_A_CONV3X3_FAST) {
            fky = ftap / 3;
            fkx = ftap - fky * 3;
            fkoff = ((int64_t)fky * p.Win + fkx) * p.Cin + fc0 + chunk * 8;
            fc0 += BK;
            if (fc0 >= p.Cin) { fc0 = 0; ++ftap; }
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (AMODE == APAD_A_CONV3X3_FAST) {
                const int iy = ra[i].oy + fky, ix = ra[i].ox + fkx;
                const bool ok = ra[i].valid && (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                const u32x4 z0 = {0u, 0u, 0u, 0u};
                gA[i] = ok ? *reinterpret_cast<const u32x4*>(p.a + (ra[i].base + fkoff) * 2) : z0;
            } else
            gA[i] = __builtin_bit_cast(u32x4, load_a<DT, AMODE>(p, ra[i], k));
            const u32x4 z = {0u, 0u, 0u, 0u};
            gB[i] = (wv[i] && k < p.K) ? *reinterpret_cast<const u32x4*>(p.w + (wb[i] + k) * 2) : z;
        }
    };
    auto sstore = [&](const u32x4* gA, const u32x4* gB, int so = 0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int rl = (tid >> 3) + 32 * i;
            *reinterpret_cast<u32x4*>(smem + gso + so + lds_off(rl, chunk)) = gA[i];
            *reinterpret_cast<u32x4*>(smem + gso + so + A_BYTES + lds_off(rl, chunk)) = gB[i];
        }
    };
    auto compute = [&](int so = 0) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ks * 2 + half;
            typename E::v8 af[MI], bf[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                af[i] = as_v8<DT>(*reinterpret_cast<const uint4*>(smem + gso + so + lds_off(wm * WT + i * 32 + l31, ch)));
                bf[i] = as_v8<DT>(
                    *reinterpret_cast<const uint4*>(smem + gso + so + A_BYTES + lds_off(wn * WT + i * 32 + l31, ch)));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = E::mfma32(af[i], bf[j], acc[i][j]);
        }
    };

    // CONV3X3_FAST tracks (tap, channel) of the next tile to load incrementally: skip n tiles
    auto fskip = [&](int n) {
        if constexpr (AMODE == APAD_A_CONV3X3_FAST) {
            for (int i = 0; i < n; ++i) {
                fc0 += BK;
                if (fc0 >= p.Cin) { fc0 = 0; ++ftap; }
            }
        }
    };
    if constexpr (KG > 1) {
    // in-workgroup split-K: group kg owns k-tiles kg, kg + KG, ...; every group runs the same number of barriers
    static_assert(KG == 1 || (NS == 1 && PF == 1), "the K-group loop is written for one LDS stage per group");
    const int nit = (nk + KG - 1) / KG;
    fskip(kg);
    if (kg < nk) {
        gload(kg, ga[0], gb[0]);
        fskip(KG - 1);
        sstore(ga[0], gb[0]);
    }
    __syncthreads();
    for (int it = 0; it < nit; ++it) {
        const int kt = it * KG + kg;
        const bool more = kt + KG < nk;  // wave-uniform
        if (more) {
            gload(kt + KG, ga[0], gb[0]);
            fskip(KG - 1);
        }
        if (kt < nk) compute();
        if (it + 1 < nit) {
            __syncthreads();
            if (more) sstore(ga[0], gb[0]);
        }
        __syncthreads();
    }
    // partial accumulators of groups 1.. -> LDS (behind the epilogue tile), summed by group 0 in group order (deterministic)
    float* red = reinterpret_cast<float*>(smem + (T::CT_BYTES + 15) / 16 * 16);
    if (kg > 0) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 4)
                    *reinterpret_cast<float4*>(&red[((((kg - 1) * MI + i) * MI + j) * 4 + (r >> 2)) * 1024 + tid * 4]) =
                        make_float4(acc[i][j][r], acc[i][j][r + 1], acc[i][j][r + 2], acc[i][j][r + 3]);
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {
                        const float4 v = *reinterpret_cast<const float4*>(&red[((((g - 1) * MI + i) * MI + j) * 4 + (r >> 2)) * 1024 + tid * 4]);
                        acc[i][j][r] += v.x; acc[i][j][r + 1] += v.y; acc[i][j][r + 2] += v.z; acc[i][j][r + 3] += v.w;
                    }
    }
    } else
    if constexpr (NS == 2) {
    // Two LDS stages: the next k-tile is written while the current one is being read -> ONE barrier per k-tile.  Pays
    // on long reductions (3x3 convolutions with K >= 2048: -3..7 %); the doubled LDS footprint halves the resident
    // workgroups, which costs more than it gains on short-K launches (K = 256, M = 64000: 22.9 -> 25.7 us), so the
    // launcher selects it per problem.
    gload(0, ga[0], gb[0]);
    sstore(ga[0], gb[0], 0);
    if (nk > 1) gload(1, ga[0], gb[0]);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = (kt & 1) * T::STAGE;
        if (kt + 1 < nk) sstore(ga[0], gb[0], cur ^ T::STAGE);  // stage last read in iteration kt-1 (barrier since)
        if (kt + 2 < nk) gload(kt + 2, ga[0], gb[0]);
        compute(cur);
        __syncthreads();
    }
    } else {
#pragma unroll
    for (int s0 = 0; s0 < PF; ++s0)
        if (s0 < nk) gload(s0, ga[s0], gb[s0]);
    sstore(ga[0], gb[0]);
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = kt0 + u;
            if (kt < nk) {  // uniform
                if (kt + PF < nk) gload(kt + PF, ga[u], gb[u]);  // set u was copied to LDS before this iteration
                compute();
                __syncthreads();
                if (kt + 1 < nk) {
                    sstore(ga[(u + 1) % PF], gb[(u + 1) % PF]);
                    __syncthreads();
                }
            }
        }
    }

    }

    // ---- epilogue: acc (+bias, +rowgroup bias, activation) -> LDS tile ----
    typename E::elem* ct = reinterpret_cast<typename E::elem*>(smem);
    int64_t step = p.step_ptr ? (int64_t)*p.step_ptr : 0;
    const bool one_group = p.rows_per_group >= p.M;  // table mode: every row reads row `step` (no 64-bit divisions)
    if (kg == 0)
#pragma unroll
    for (int j = 0; j < MI; ++j) {
        const int nl = wn * WT + j * 32 + l31;
        const bool nvalid = wrow_valid(nl);
        const int64_t wr = nvalid ? wrow(nl) : 0;
        const float bv = (p.bias && nvalid) ? ld_elem<DT>(p.bias, wr) : 0.f;
        const float rg0 = (p.rg && one_group && nvalid) ? ld_elem<DT>(p.rg, step * p.ld_rg + wr) : 0.f;
        const bool lnf = p.rs_in != nullptr;
        const float lcs = (lnf && nvalid) ? p.ln_cs[wr] : 0.f, lbb = (lnf && nvalid) ? p.ln_bb[wr] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * WT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[i][j][r] + bv + rg0;
                if (lnf) v = rstat[ml][1] * (acc[i][j][r] - rstat[ml][0] * lcs) + lbb + rg0;
                if (p.rg && !one_group) {
                    int64_t m = m0 + ml;
                    if (m < p.M && nvalid) v += ld_elem<DT>(p.rg, (m / p.rows_per_group + step) * p.ld_rg + wr);
                }
                if (EPI == APAD_EPI_SILU) v = silu_f(v);
                if (EPI == APAD_EPI_GELU) v = gelu_erf_f(v);
                if (EPI == APAD_EPI_TANH) v = tanhf(v);
                ct[ml * C_LD + nl] = (typename E::elem)v;
            }
        }
    }
    __syncthreads();

    // fused q|k|v: the tile lies in exactly one third of the columns (C % tile == 0, checked on the host)
    const int Cq = (int)(p.N / 3);
    const int qseg = (OUTMODE == APAD_OUT_QKV) ? (int)(n0 / Cq) : 0;
    if (OUTMODE == APAD_OUT_ROWMAJOR || (OUTMODE == APAD_OUT_QKV && qseg < 2)) {
        uint8_t* const obase = (OUTMODE == APAD_OUT_QKV && qseg == 1) ? p.out2 : p.out;
        const int64_t ncol0 = (OUTMODE == APAD_OUT_QKV) ? (int64_t)qseg * Cq : 0;
        constexpr int VPR = BN_OUT / 8;  // 16-byte vectors per output row
        if (p.rs_out != nullptr && VPR >= 8 && OUTMODE == APAD_OUT_ROWMAJOR) {
            // the same store loop, plus the row statistics of what is stored: 8 consecutive lanes own 64 consecutive columns of one
            // row (BM * VPR is a multiple of the thread count, so a group is never split and every lane takes part in the shuffles)
            for (int idx = threadIdx.x; idx < BM * VPR; idx += NT) {
                const int rl = idx / VPR, vc = idx - rl * VPR;
                const int64_t m = m0 + rl, n = n0 + vc * 8;
                const bool ok = m < p.M && n < p.N;
                float f[8];
                unpack8<DT>(*reinterpret_cast<const uint4*>(&ct[rl * C_LD + vc * 8]), f);
                if (p.residual && ok) {
                    float rr[8];
                    const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
                    unpack8<DT>(*reinterpret_cast<const uint4*>(p.residual + (rm * p.ldr + n) * 2), rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)(typename E::elem)f[e] + rr[e];
                }
                const uint4 pk = pack8<DT>(f);
                float s1 = 0.f, s2 = 0.f;
                if (ok) {
                    *reinterpret_cast<uint4*>(obase + (m * p.ldo + (n - ncol0)) * 2) = pk;
                    float g[8];
                    unpack8<DT>(pk, g);  // statistics of the ROUNDED values: what the consumer will read
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1 += g[e];
                        s2 = __builtin_fmaf(g[e], g[e], s2);
                    }
                }
#pragma unroll
                for (int o_ = 1; o_ < 8; o_ <<= 1) {
                    s1 += __shfl_xor(s1, o_);
                    s2 += __shfl_xor(s2, o_);
                }
                if (ok && (vc & 7) == 0) {
                    float* dst = p.rs_out + (m * p.rs_out_tiles + (n >> 6)) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        } else
        for (int idx = threadIdx.x; idx < BM * VPR; idx += NT) {
            const int rl = idx / VPR, vc = idx - rl * VPR;
            const int64_t m = m0 + rl, n = n0 + vc * 8;
            if (m >= p.M || n >= p.N) continue;
            float f[8];
            unpack8<DT>(*reinterpret_cast<const uint4*>(&ct[rl * C_LD + vc * 8]), f);
            if (EPI == APAD_EPI_GEGLU) {
                float g[8];
                unpack8<DT>(*reinterpret_cast<const uint4*>(&ct[rl * C_LD + GH + vc * 8]), g);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const apad_f32x2 ge = gelu_erf_2((apad_f32x2){g[e], g[e + 1]});
                    f[e] *= ge[0];
                    f[e + 1] *= ge[1];
                }
            }
            if (p.residual) {
                float rr[8];
                const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
                unpack8<DT>(*reinterpret_cast<const uint4*>(p.residual + (rm * p.ldr + n) * 2), rr);
                // the un-fused reference rounds the linear output to the storage type before the add
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)(typename E::elem)f[e] + rr[e];
            }
            *reinterpret_cast<uint4*>(obase + (m * p.ldo + (n - ncol0)) * 2) = pack8<DT>(f);
        }
    } else {  // APAD_OUT_VT (or the v third of APAD_OUT_QKV): consecutive lanes -> consecutive tokens of one (head, dd) row
        typename E::elem* o = reinterpret_cast<typename E::elem*>(OUTMODE == APAD_OUT_QKV ? p.out3 : p.out);
        const int64_t nsub = (OUTMODE == APAD_OUT_QKV) ? 2 * (int64_t)Cq : 0;
        if (OUTMODE == APAD_OUT_QKV && p.out4 != nullptr) {  // v row-major as well (the training step keeps both forms)
            for (int idx = threadIdx.x; idx < BM * (BN / 8); idx += NT) {
                const int rl = idx / (BN / 8), vc = idx - rl * (BN / 8);
                const int64_t m = m0 + rl, n = n0 + vc * 8;
                if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(p.out4 + (m * p.ldo + (n - nsub)) * 2) = *reinterpret_cast<const uint4*>(&ct[rl * C_LD + vc * 8]);
            }
        }
        for (int idx = threadIdx.x; idx < BM * BN; idx += NT) {
            const int nl = idx / BM, rl = idx % BM;
            const int64_t m = m0 + rl;
            int64_t n = n0 + nl;
            if (m >= p.M || n >= p.N) continue;
            n -= nsub;
            const int64_t b = m / p.L;
            const int l = (int)(m - b * p.L);
            const int h = (int)(n / p.head_dim), dd = (int)(n - (int64_t)h * p.head_dim);
            o[((b * p.heads + h) * p.head_dim + dd) * p.Lpad + l] = ct[rl * C_LD + nl];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LDS-DMA ring form of the 64x64 tile for latency-bound plain launches (round 3).  The tiled kernel above keeps one k-tile per K group
// in flight -- global load -> registers -> LDS -> barrier -- so a launch of a few hundred workgroups with 4 .. 40 k-tiles each is a chain
// of exposed load latencies.  Here both operands go HBM / L2 -> LDS by `buffer_load ... lds` into a FOUR-stage ring (three k-tiles in
// flight per workgroup, two workgroups per CU), one barrier per k-tile, fragment reads in inline asm (see cgemm.hip for why).  Same LDS
// layout (lds_off) as the tiled kernel, produced on the source side.  KSUM = 2 reproduces the K-group summation of the tiled kernel
// for the shapes its (N, K) rule selects -- even k-tiles into one accumulator, odd ones into another, then even + odd -- so that a
// row's result does not depend on which kernel ran; KSUM = 1: k-tiles in order.  Epilogue: the tiled kernel's, at MI = 1.
// Where it is used: the TRAINING step (apad_set_gemm_ring(1), AdapterTrainer: ~2600 launches of <= 4000 rows per step on one stream);
// in the denoise step the 64-token level runs on two streams and the ring's 66 KB of LDS per workgroup (tiled: 32 KB) costs their
// co-residency: measured slower there (44.54 -> 44.83 ms), so inference keeps the tiled kernel.
typedef __attribute__((address_space(3))) void* g_lds_ptr;
constexpr int RSTAGE = 2 * 64 * BK * 2, RNST = 4, RSMEM = RNST * RSTAGE + 64 * 2 * 4;  // 16 384 per stage + the row statistics
constexpr uint32_t R_OOB = 0x80000000u;

template <int DT, int EPI, int OUTMODE, int KSUM>
__global__ __launch_bounds__(256) void gemm_ring_kernel(GemmP p, uint32_t a_bytes, uint32_t a2_bytes, uint32_t w_bytes) {
    constexpr int TM = 64, BM = 64, BN = 64, C_LD = TM + 8, WT = 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float (*rstat)[2] = reinterpret_cast<float (*)[2]>(smem + RNST * RSTAGE);
    using E = ET<DT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    constexpr int BN_OUT = (EPI == APAD_EPI_GEGLU) ? BN / 2 : BN;
    constexpr int GH = BN / 2;
    int mt, nt;
    {
        const int nN = p.n_tiles, nM = p.m_tiles;
        const int b = blockIdx.x;
        const int full = (nM / 8) * 8 * nN;
        if (b < full) {
            const int grp = b / (8 * nN), rem = b - grp * 8 * nN;
            nt = rem >> 3;
            mt = grp * 8 + (rem & 7);
        } else {
            const int rem = b - full, tail = nM - (nM / 8) * 8;
            nt = rem / tail;
            mt = (nM / 8) * 8 + rem - nt * tail;
        }
    }
    const int64_t m0 = (int64_t)mt * BM;
    const int64_t n0 = (int64_t)nt * BN_OUT;
    auto wrow = [&](int nl) -> int64_t {
        if (EPI == APAD_EPI_GEGLU) return nl < GH ? n0 + nl : p.N + n0 + (nl - GH);
        return n0 + nl;
    };
    if (p.rs_in != nullptr && tid < BM) {
        const int64_t m = m0 + tid;
        float s1 = 0.f, s2 = 0.f;
        if (m < p.M) {
            const float* src = p.rs_in + m * p.rs_in_tiles * 2;
            for (int t_ = 0; t_ < p.rs_in_tiles; ++t_) {
                s1 += src[2 * t_];
                s2 += src[2 * t_ + 1];
            }
        }
        const float mean = s1 / (float)p.K;
        const float var = fmaxf(s2 / (float)p.K - mean * mean, 0.f);
        rstat[tid][0] = mean;
        rstat[tid][1] = rsqrtf(var + p.ln_eps);
    }
    // ---- DMA sources: wave w fills blocks 2w, 2w + 1 (8 rows each) of the A tile and of the W tile; lane -> (row, LDS slot), the
    //      slot holds source chunk slot ^ ((row >> 1) & 7) = lds_off's swizzle ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.w), 0, (int)w_bytes, 0x00020000);
    uint32_t aoff[2], aoff2[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (wave * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        const int64_t m = m0 + R;
        const bool valid = m < p.M;
        aoff[i] = valid ? (uint32_t)((p.a_mod > 0 ? m % p.a_mod : m) * p.lda * 2 + c * 16) : R_OOB;
        aoff2[i] = (valid && p.a2 != nullptr) ? (uint32_t)((p.a2_mod > 0 ? m % p.a2_mod : m) * p.lda2 * 2 + c * 16) : R_OOB;
        boff[i] = (uint32_t)(wrow(R) * p.ldw * 2 + c * 16);
    }
    const int nk = (int)(p.K / BK);
    auto request = [&](int kt, int stage) {
        uint8_t* st = smem + stage * RSTAGE;
        const bool second = p.a2 != nullptr && kt * BK >= p.ksplit;  // wave-uniform
        const int soff = (second ? kt * BK - p.ksplit : kt * BK) * 2;
        // (the descriptor is rebuilt from scalar selects: two descriptors selected per call were kept in scratch)
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(second ? p.a2 : p.a), 0,
                                                                            (int)(second ? a2_bytes : a_bytes), 0x00020000);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (g_lds_ptr)(st + (wave * 2 + i) * 1024), 16, second ? aoff2[i] : aoff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (g_lds_ptr)(st + 64 * BK * 2 + (wave * 2 + i) * 1024), 16, boff[i], kt * (BK * 2), 0, 0);
    };
    uint32_t fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = (uint32_t)(l31 * 128 + (((ks * 2 + half) ^ ((l31 >> 1) & 7)) << 4));
    const uint32_t lds0 = (uint32_t)(size_t)(g_lds_ptr)smem;
    const uint32_t abase = lds0 + (uint32_t)(wm * 32 * 128), bbase = lds0 + (uint32_t)(64 * BK * 2 + wn * 32 * 128);
    f32x16 acc, acc1;  // (named, not an array: a lambda-captured one-element array of vectors went to scratch)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (KSUM == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
    }

#pragma unroll
    for (int t = 0; t < RNST - 1; ++t)
        if (t < nk) request(t, t);
    auto ktile = [&](int t, auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value;
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone's pieces of tile t are in LDS, everyone is past its reads of tile t - 1
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (t + RNST - 1 < nk) request(t + RNST - 1, (S + RNST - 1) % RNST);
        u32x4 fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t aa = abase + (uint32_t)(S * RSTAGE) + fo[ks], bb = bbase + (uint32_t)(S * RSTAGE) + fo[ks];
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[ks]) : "v"(bb));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // (RNST = 4 is even: the stage index parity is the k-tile parity, i.e. the K group of the tiled kernel)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if constexpr (KSUM == 2 && (S & 1) == 1)
                acc1 = E::mfma32(__builtin_bit_cast(typename E::v8, fa[ks]), __builtin_bit_cast(typename E::v8, fb[ks]), acc1);
            else
                acc = E::mfma32(__builtin_bit_cast(typename E::v8, fa[ks]), __builtin_bit_cast(typename E::v8, fb[ks]), acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int t = 0; t < nk; t += 4) {
        ktile(t, std::integral_constant<int, 0>{});
        if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < nk) ktile(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < nk) ktile(t + 3, std::integral_constant<int, 3>{});
    }
    if constexpr (KSUM == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc1[r];  // group 0 + group 1, as the tiled kernel's reduction
    }
    __syncthreads();  // the stages are dead (rstat lives behind them)

    // ---- epilogue: the tiled kernel's (gemm_kernel), one MFMA tile per wave ----
    typename E::elem* ct = reinterpret_cast<typename E::elem*>(smem);
    const int64_t step = p.step_ptr ? (int64_t)*p.step_ptr : 0;
    const bool one_group = p.rows_per_group >= p.M;
    {
        const int nl = wn * WT + l31;
        const int64_t wr = wrow(nl);
        const float bv = p.bias ? ld_elem<DT>(p.bias, wr) : 0.f;
        const float rg0 = (p.rg && one_group) ? ld_elem<DT>(p.rg, step * p.ld_rg + wr) : 0.f;
        const bool lnf = p.rs_in != nullptr;
        const float lcs = lnf ? p.ln_cs[wr] : 0.f, lbb = lnf ? p.ln_bb[wr] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * WT + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[r] + bv + rg0;
            if (lnf) v = rstat[ml][1] * (acc[r] - rstat[ml][0] * lcs) + lbb + rg0;
            if (p.rg && !one_group) {
                const int64_t m = m0 + ml;
                if (m < p.M) v += ld_elem<DT>(p.rg, (m / p.rows_per_group + step) * p.ld_rg + wr);
            }
            if (EPI == APAD_EPI_SILU) v = silu_f(v);
            if (EPI == APAD_EPI_GELU) v = gelu_erf_f(v);
            ct[ml * C_LD + nl] = (typename E::elem)v;
        }
    }
    __syncthreads();
    const int Cq = (int)(p.N / 3);
    const int qseg = (OUTMODE == APAD_OUT_QKV) ? (int)(n0 / Cq) : 0;
    if (OUTMODE == APAD_OUT_ROWMAJOR || (OUTMODE == APAD_OUT_QKV && qseg < 2)) {
        uint8_t* const obase = (OUTMODE == APAD_OUT_QKV && qseg == 1) ? p.out2 : p.out;
        const int64_t ncol0 = (OUTMODE == APAD_OUT_QKV) ? (int64_t)qseg * Cq : 0;
        constexpr int VPR = BN_OUT / 8;
        if (p.rs_out != nullptr && VPR >= 8 && OUTMODE == APAD_OUT_ROWMAJOR) {
            for (int idx = tid; idx < BM * VPR; idx += 256) {
                const int rl = idx / VPR, vc = idx - rl * VPR;
                const int64_t m = m0 + rl, n = n0 + vc * 8;
                const bool ok = m < p.M && n < p.N;
                float f[8];
                unpack8<DT>(*reinterpret_cast<const uint4*>(&ct[rl * C_LD + vc * 8]), f);
                if (p.residual && ok) {
                    float rr[8];
                    const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
                    unpack8<DT>(*reinterpret_cast<const uint4*>(p.residual + (rm * p.ldr + n) * 2), rr);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = (float)(typename E::elem)f[e] + rr[e];
                }
                const uint4 pk = pack8<DT>(f);
                float s1 = 0.f, s2 = 0.f;
                if (ok) {
                    *reinterpret_cast<uint4*>(obase + (m * p.ldo + (n - ncol0)) * 2) = pk;
                    float g[8];
                    unpack8<DT>(pk, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1 += g[e];
                        s2 = __builtin_fmaf(g[e], g[e], s2);
                    }
                }
#pragma unroll
                for (int o_ = 1; o_ < 8; o_ <<= 1) {
                    s1 += __shfl_xor(s1, o_);
                    s2 += __shfl_xor(s2, o_);
                }
                if (ok && (vc & 7) == 0) {
                    float* dst = p.rs_out + (m * p.rs_out_tiles + (n >> 6)) * 2;
                    dst[0] = s1;
                    dst[1] = s2;
                }
            }
        } else
        for (int idx = tid; idx < BM * VPR; idx += 256) {
            const int rl = idx / VPR, vc = idx - rl * VPR;
            const int64_t m = m0 + rl, n = n0 + vc * 8;
            if (m >= p.M || n >= p.N) continue;
            float f[8];
            unpack8<DT>(*reinterpret_cast<const uint4*>(&ct[rl * C_LD + vc * 8]), f);
            if (EPI == APAD_EPI_GEGLU) {
                float g[8];
                unpack8<DT>(*reinterpret_cast<const uint4*>(&ct[rl * C_LD + GH + vc * 8]), g);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const apad_f32x2 ge = gelu_erf_2((apad_f32x2){g[e], g[e + 1]});
                    f[e] *= ge[0];
                    f[e + 1] *= ge[1];
                }
            }
            if (p.residual) {
                float rr[8];
                const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
                unpack8<DT>(*reinterpret_cast<const uint4*>(p.residual + (rm * p.ldr + n) * 2), rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (float)(typename E::elem)f[e] + rr[e];
            }
            *reinterpret_cast<uint4*>(obase + (m * p.ldo + (n - ncol0)) * 2) = pack8<DT>(f);
        }
    } else {
        typename E::elem* o = reinterpret_cast<typename E::elem*>(OUTMODE == APAD_OUT_QKV ? p.out3 : p.out);
        const int64_t nsub = (OUTMODE == APAD_OUT_QKV) ? 2 * (int64_t)Cq : 0;
        if (OUTMODE == APAD_OUT_QKV && p.out4 != nullptr) {
            for (int idx = tid; idx < BM * (BN / 8); idx += 256) {
                const int rl = idx / (BN / 8), vc = idx - rl * (BN / 8);
                const int64_t m = m0 + rl, n = n0 + vc * 8;
                if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(p.out4 + (m * p.ldo + (n - nsub)) * 2) = *reinterpret_cast<const uint4*>(&ct[rl * C_LD + vc * 8]);
            }
        }
        for (int idx = tid; idx < BM * BN; idx += 256) {
            const int nl = idx / BM, rl = idx % BM;
            const int64_t m = m0 + rl;
            int64_t n = n0 + nl;
            if (m >= p.M || n >= p.N) continue;
            n -= nsub;
            const int64_t b = m / p.L;
            const int l = (int)(m - b * p.L);
            const int h = (int)(n / p.head_dim), dd = (int)(n - (int64_t)h * p.head_dim);
            o[((b * p.heads + h) * p.head_dim + dd) * p.Lpad + l] = ct[rl * C_LD + nl];
        }
    }
}

// the ring kernel's envelope: plain A, K % 64 == 0, all weight rows of the tiles in range, operands below 2 GB; returns 1 when it
// does not apply
template <int DT, int EPI, int OUTMODE>
int launch_ring(const GemmP& p, bool kgroups, hipStream_t s) {
    constexpr int BN_OUT = (EPI == APAD_EPI_GEGLU) ? 32 : 64;
    if (p.K % 64 != 0 || p.N % BN_OUT != 0 || p.lda % 8 != 0 || p.M >= (1LL << 30)) return 1;
    const int64_t rows_a = p.a_mod > 0 ? p.a_mod : p.M;
    const int64_t a_bytes = ((rows_a - 1) * p.lda + (p.a2 ? p.ksplit : p.K)) * 2;
    const int64_t a2_bytes = p.a2 ? (((p.a2_mod > 0 ? p.a2_mod : p.M) - 1) * p.lda2 + (p.K - p.ksplit)) * 2 : 0;
    const int64_t w_bytes = ((int64_t)(p.wrows - 1) * p.ldw + p.K) * 2;
    if (a_bytes >= (1LL << 31) || a2_bytes >= (1LL << 31) || w_bytes >= (1LL << 31)) return 1;
    GemmP q = p;
    q.n_tiles = (int)(p.N / BN_OUT);
    q.m_tiles = (int)((p.M + 63) / 64);
    dim3 grid((unsigned)(q.n_tiles * q.m_tiles));
    auto go = [&](auto kern) {
        static unsigned devs = 0;
        if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), RSMEM, &devs) != 0) return -1;
        hipLaunchKernelGGL(kern, grid, dim3(256), RSMEM, s, q, (uint32_t)a_bytes, (uint32_t)a2_bytes, (uint32_t)w_bytes);
        return apad_check_launch("apad_gemm(ring)");
    };
    if (kgroups) return go(gemm_ring_kernel<DT, EPI, OUTMODE, 2>);
    return go(gemm_ring_kernel<DT, EPI, OUTMODE, 1>);
}

// process-wide switch of the ring form (apad_set_gemm_ring): -1 = the APAD_GEMM_RING environment variable (default 0)
int g_ring_mode = -1;

template <int DT, int AMODE, int EPI, int OUTMODE, int TM, int NS = 1, int KG = 1>
int launch_tm(const GemmP& p, hipStream_t s) {
    constexpr int BN_OUT = (EPI == APAD_EPI_GEGLU) ? TM / 2 : TM;
    GemmP q = p;
    q.n_tiles = (int)((p.N + BN_OUT - 1) / BN_OUT);
    q.m_tiles = (int)((p.M + TM - 1) / TM);
    dim3 grid((unsigned)(q.n_tiles * q.m_tiles));
    hipLaunchKernelGGL((gemm_kernel<DT, AMODE, EPI, OUTMODE, TM, NS, KG>), grid, dim3(256 * KG), 0, s, q);
    return apad_check_launch("apad_gemm");
}

template <int DT, int AMODE, int EPI, int OUTMODE>
int launch(const GemmP& p, hipStream_t s) {
    // 128-tiles unless they would leave the chip under-filled (< ~2 workgroups per CU)
    constexpr int BN_OUT = (EPI == APAD_EPI_GEGLU) ? 64 : 128;
    const int64_t blocks128 = ((p.N + BN_OUT - 1) / BN_OUT) * ((p.M + 127) / 128);
    // long reductions amortise the under-fill: with K >= 1024 the 128-tile wins from ~1.25 workgroups per CU (measured:
    // conv 63x4 384->384 76.5 -> 71.5 us, FF2 M=16128 K=1536 38.5 -> 37.1 us), short-K launches prefer the 64-tile
    constexpr int t128_min = 512;  // (A/B knob)
    const bool t128 = blocks128 >= t128_min || (blocks128 >= 320 && t128_min <= 512 && p.K >= 1024);
    if constexpr (AMODE == APAD_A_CONV3X3_FAST && EPI == APAD_EPI_NONE && OUTMODE == APAD_OUT_ROWMAJOR) {
        // under-filled 3x3 convolutions (the 640- / 384-wide resnets: 90..180 k-tiles on a few hundred 64x64 tiles): K groups
        constexpr int conv_kg = 0;  // (A/B knob)
        if (!t128 && conv_kg >= 4) return launch_tm<DT, AMODE, EPI, OUTMODE, 64, 1, 4>(p, s);
        if (!t128 && conv_kg >= 2) return launch_tm<DT, AMODE, EPI, OUTMODE, 64, 1, 2>(p, s);
    }
    if constexpr (AMODE == APAD_A_CONV3X3_FAST || AMODE == APAD_A_CONV3X3 || AMODE == APAD_A_CONV1D) {
        constexpr bool one_stage = false;
        // long reductions on launches of <= ~4 workgroups per CU: two LDS stages, one barrier per k-tile (larger grids
        // lose more from the halved residency than they gain: 250x16 128->128 118.9 -> 132.6 us)
        if (p.K >= 2048 && blocks128 <= 1024 && !one_stage)
            return t128 ? launch_tm<DT, AMODE, EPI, OUTMODE, 128, 2>(p, s) : launch_tm<DT, AMODE, EPI, OUTMODE, 64, 2>(p, s);
    }
    if constexpr (AMODE == APAD_A_PLAIN && (EPI == APAD_EPI_NONE || EPI == APAD_EPI_GEGLU) && (OUTMODE == APAD_OUT_ROWMAJOR || OUTMODE == APAD_OUT_QKV) &&
                  !(EPI == APAD_EPI_GEGLU && OUTMODE == APAD_OUT_QKV)) {
        // latency-bound launches: the LDS-DMA ring form (apad_set_gemm_ring / APAD_GEMM_RING: 0 off, 1 = grids the 128-tile rule calls
        // under-filled, 2 = every launch; below APAD_GEMM_RING_MAX_M = 16000 rows).  Bit-equal to the tiled kernels (same k-summation order, K groups included).
        static const int ring_env = [] { const char* e = getenv("APAD_GEMM_RING"); return e ? atoi(e) : 0; }();
        constexpr int kg_mode_r = 2;
        constexpr int ring_max_m = 16000;
        const int ring_mode = g_ring_mode >= 0 ? g_ring_mode : ring_env;
        constexpr int ring_max_wg = (1 << 30);
        constexpr int ring_min_wg = 0;
        const int64_t ring_wgs = ((p.M + 63) / 64) * (p.N / (EPI == APAD_EPI_GEGLU ? 32 : 64));
        if (ring_mode && p.K >= 128 && p.M < ring_max_m && (!t128 || ring_mode >= 2) && ring_wgs <= ring_max_wg && ring_wgs >= ring_min_wg) {
            constexpr int ring_kg = 1;
            const bool kgroups = EPI == APAD_EPI_NONE && kg_mode_r >= 2 && p.K >= 384 && p.N >= 640;
            if (!kgroups || ring_kg) {
                const int rc = launch_ring<DT, EPI, OUTMODE>(p, kgroups, s);
                if (rc <= 0) return rc;
            }
        }
    }
    if constexpr (AMODE == APAD_A_PLAIN && EPI == APAD_EPI_NONE) {
        // K groups inside the workgroup for the skinny launches of the 640- / 384-wide levels.  The choice depends on (N, K) ONLY,
        // never on M, and both tile sizes implement it: a row's k-summation order must not change with the batch size -- a clip's
        // result is bit-identical whatever batch it rides in (tests/test_gpu_unet.py::test_full_size_clips_are_independent_of_their_batch)
        constexpr int kg_mode = 2;  // (A/B knob: 1 = off)
        // (N >= 640: the 640-wide level's to_q / to_out / FF2 / q|k|v.  At N = 384 the FF2 of the 384-wide level, M = 16128 on 128-tiles,
        //  measured 39 -> 58 us with K groups: the rule stops short of it)
        if (kg_mode >= 2 && p.K >= 384 && p.N >= 640)
            return t128 ? launch_tm<DT, AMODE, EPI, OUTMODE, 128, 1, 2>(p, s) : launch_tm<DT, AMODE, EPI, OUTMODE, 64, 1, 2>(p, s);
    }
    if (t128) return launch_tm<DT, AMODE, EPI, OUTMODE, 128>(p, s);
    return launch_tm<DT, AMODE, EPI, OUTMODE, 64>(p, s);
}

template <int DT, int AMODE>
int dispatch_epi(const GemmP& p, int epi, int outmode, hipStream_t s) {
    if (outmode == APAD_OUT_VT) {
        APAD_CHECK(epi == APAD_EPI_NONE, "apad_gemm: APAD_OUT_VT supports epilogue NONE only");
        return launch<DT, AMODE, APAD_EPI_NONE, APAD_OUT_VT>(p, s);
    }
    if (outmode == APAD_OUT_QKV) {
        APAD_CHECK(epi == APAD_EPI_NONE && AMODE == APAD_A_PLAIN, "apad_gemm: APAD_OUT_QKV supports plain A / epilogue NONE only");
        if constexpr (AMODE == APAD_A_PLAIN) return launch<DT, AMODE, APAD_EPI_NONE, APAD_OUT_QKV>(p, s);
    }
    switch (epi) {
        case APAD_EPI_NONE: return launch<DT, AMODE, APAD_EPI_NONE, APAD_OUT_ROWMAJOR>(p, s);
        case APAD_EPI_SILU: return launch<DT, AMODE, APAD_EPI_SILU, APAD_OUT_ROWMAJOR>(p, s);
        case APAD_EPI_GELU: return launch<DT, AMODE, APAD_EPI_GELU, APAD_OUT_ROWMAJOR>(p, s);
        case APAD_EPI_GEGLU: return launch<DT, AMODE, APAD_EPI_GEGLU, APAD_OUT_ROWMAJOR>(p, s);
    }
    if (epi == APAD_EPI_TANH || epi == APAD_EPI_RELU || epi == APAD_EPI_GELU_TANH || epi == APAD_EPI_GEGLU_TANH) {
        apad_set_error("apad_gemm: epilogue %d is an fp32-mode epilogue (dtype APAD_F32; 16-bit: tanh in conv1d mode only)", epi);
        return -1;
    }
    apad_set_error("apad_gemm: unknown epilogue %d", epi);
    return -1;
}

template <int DT> int dispatch_amode(const GemmP& p, const apad_gemm_desc* d, hipStream_t s) {
    switch (d->a_mode) {
        case APAD_A_PLAIN: return dispatch_epi<DT, APAD_A_PLAIN>(p, d->epilogue, d->out_mode, s);
        case APAD_A_CONV3X3:
            APAD_CHECK(d->epilogue == APAD_EPI_NONE && d->out_mode == APAD_OUT_ROWMAJOR,
                       "apad_gemm: conv3x3 supports epilogue NONE / row-major output only");
            if (d->Cin % 64 == 0 && d->Hup == 0)
                return launch<DT, APAD_A_CONV3X3_FAST, APAD_EPI_NONE, APAD_OUT_ROWMAJOR>(p, s);
            return launch<DT, APAD_A_CONV3X3, APAD_EPI_NONE, APAD_OUT_ROWMAJOR>(p, s);
        case APAD_A_PATCH16:
            APAD_CHECK(d->epilogue == APAD_EPI_NONE && d->out_mode == APAD_OUT_ROWMAJOR,
                       "apad_gemm: patch16 supports epilogue NONE / row-major output only");
            return launch<DT, APAD_A_PATCH16, APAD_EPI_NONE, APAD_OUT_ROWMAJOR>(p, s);
        case APAD_A_CONV1D:
            APAD_CHECK((d->epilogue == APAD_EPI_NONE || d->epilogue == APAD_EPI_TANH) && d->out_mode == APAD_OUT_ROWMAJOR,
                       "apad_gemm: conv1d supports epilogue NONE / TANH and row-major output only");
            if (d->epilogue == APAD_EPI_TANH) return launch<DT, APAD_A_CONV1D, APAD_EPI_TANH, APAD_OUT_ROWMAJOR>(p, s);
            return launch<DT, APAD_A_CONV1D, APAD_EPI_NONE, APAD_OUT_ROWMAJOR>(p, s);
    }
    apad_set_error("apad_gemm: unknown a_mode %d", d->a_mode);
    return -1;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int apad_set_gemm_ring(int32_t mode) {
    const int old = g_ring_mode;
    g_ring_mode = mode;
    return old;
}

extern "C" int apad_gemm(const apad_gemm_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_gemm: null descriptor");
    if (d->dtype == APAD_F32) return apad_f32_gemm(d, (hipStream_t)stream);  // fp32 precision mode (f32_ops.hip)
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_gemm: dtype %d not supported (bf16/f16/f32)", d->dtype);
    APAD_CHECK(d->a && d->w && d->out, "apad_gemm: null operand");
    APAD_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "apad_gemm: empty problem M=%lld N=%lld K=%lld", (long long)d->M,
               (long long)d->N, (long long)d->K);
    APAD_CHECK(d->K % 8 == 0 && d->ldw % 8 == 0, "apad_gemm: K and ldw must be multiples of 8 (K=%lld ldw=%lld)",
               (long long)d->K, (long long)d->ldw);
    APAD_CHECK(al16(d->a) && al16(d->w) && al16(d->out) && al16(d->residual), "apad_gemm: pointers must be 16-byte aligned");
    GemmP p;
    p.a = (const uint8_t*)d->a;
    p.w = (const uint8_t*)d->w;
    p.out = (uint8_t*)d->out;
    p.out2 = (uint8_t*)d->out2;
    p.out3 = (uint8_t*)d->out3;
    p.out4 = (uint8_t*)d->out4;
    p.bias = (const uint8_t*)d->bias;
    p.residual = (const uint8_t*)d->residual;
    p.rg = (const uint8_t*)d->rowgroup_bias;
    p.step_ptr = d->step_ptr;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldw = d->ldw; p.ldo = d->ldo; p.ldr = d->ldr; p.ld_rg = d->ld_rg;
    p.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1;
    p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Hout = d->Hout; p.Wout = d->Wout;
    p.stride = d->stride; p.Hup = d->Hup; p.Wup = d->Wup; p.src_batch_mod = d->src_batch_mod; p.res_mod = d->residual_row_mod;
    p.heads = d->heads; p.head_dim = d->head_dim; p.L = d->L; p.Lpad = d->Lpad;
    p.wrows = (int32_t)(d->epilogue == APAD_EPI_GEGLU ? 2 * d->N : d->N);
    if (d->a_mode == APAD_A_PLAIN) {
        APAD_CHECK(d->lda % 8 == 0, "apad_gemm: lda must be a multiple of 8");
    } else if (d->a_mode == APAD_A_CONV3X3) {
        APAD_CHECK(d->Cin > 0 && d->Cin % 8 == 0 && d->K == 9LL * d->Cin, "apad_gemm: conv3x3 needs Cin%%8==0 and K==9*Cin");
        APAD_CHECK(d->stride == 1 || d->stride == 2, "apad_gemm: conv stride must be 1 or 2");
        APAD_CHECK(d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0 && d->M % ((int64_t)d->Hout * d->Wout) == 0,
                   "apad_gemm: conv geometry inconsistent with M");
        APAD_CHECK((d->Hup > 0) == (d->Wup > 0), "apad_gemm: Hup/Wup must both be set or both 0");
    } else if (d->a_mode == APAD_A_PATCH16) {
        APAD_CHECK(d->K == 256 && d->Hin % 16 == 0 && d->Win % 16 == 0, "apad_gemm: patch16 needs K==256 and H,W %% 16 == 0");
        APAD_CHECK(d->M % ((int64_t)(d->Hin / 16) * (d->Win / 16)) == 0, "apad_gemm: patch16 M inconsistent");
    } else if (d->a_mode == APAD_A_CONV1D) {
        APAD_CHECK(d->Cin > 0 && d->Cin % 8 == 0 && d->taps > 0 && d->K == (int64_t)d->taps * d->Cin, "apad_gemm: conv1d needs Cin%%8==0 and K==taps*Cin");
        APAD_CHECK(d->Hin > 0 && d->Hout > 0 && d->M % d->Hout == 0 && d->pad >= 0, "apad_gemm: conv1d geometry inconsistent with M");
        APAD_CHECK(d->transposed ? d->stride >= 1 : d->dilation >= 1, "apad_gemm: conv1d needs dilation >= 1 (stride >= 1 when transposed)");
    }
    p.taps = d->taps; p.dilation = d->dilation; p.pad = d->pad; p.transposed = d->transposed; p.pre_act = d->a_pre_act;
    p.pre_slope = d->a_pre_slope;
    p.lead = d->conv_asym_pad ? 0 : 1;
    if (d->out_mode == APAD_OUT_ROWMAJOR) {
        APAD_CHECK(d->N % 8 == 0 && d->ldo % 8 == 0, "apad_gemm: N and ldo must be multiples of 8");
        if (d->residual) APAD_CHECK(d->ldr % 8 == 0, "apad_gemm: ldr must be a multiple of 8");
        if (d->epilogue == APAD_EPI_GEGLU) APAD_CHECK(d->N % 64 == 0, "apad_gemm: GEGLU needs N %% 64 == 0");
    } else if (d->out_mode == APAD_OUT_QKV) {
        APAD_CHECK(d->out2 && d->out3 && al16(d->out2) && al16(d->out3), "apad_gemm: APAD_OUT_QKV needs 16-byte aligned out2 / out3");
        APAD_CHECK(al16(d->out4), "apad_gemm: out4 must be 16-byte aligned");
        APAD_CHECK(d->heads > 0 && d->head_dim > 0 && d->L > 0 && d->Lpad >= d->L && d->N == 3LL * d->heads * d->head_dim &&
                       d->M % d->L == 0 && (d->N / 3) % 128 == 0 && d->ldo % 8 == 0,
                   "apad_gemm: fused q|k|v geometry inconsistent (needs C %% 128 == 0)");
        APAD_CHECK(!d->residual, "apad_gemm: fused q|k|v takes no residual");
    } else if (d->out_mode == APAD_OUT_VT) {
        APAD_CHECK(d->heads > 0 && d->head_dim > 0 && d->L > 0 && d->Lpad >= d->L && d->N == (int64_t)d->heads * d->head_dim &&
                       d->M % d->L == 0,
                   "apad_gemm: V^T output geometry inconsistent");
        APAD_CHECK(!d->residual, "apad_gemm: V^T output takes no residual");
    } else {
        apad_set_error("apad_gemm: unknown out_mode %d", d->out_mode);
        return -1;
    }
    if (d->rowgroup_bias) APAD_CHECK(d->ld_rg > 0, "apad_gemm: rowgroup_bias needs ld_rg");
    p.a2 = (const uint8_t*)d->a2; p.lda2 = d->lda2; p.ksplit = d->k_split; p.a_mod = d->a_row_mod; p.a2_mod = d->a2_row_mod;
    if (d->a2 != nullptr)
        APAD_CHECK(d->a_mode == APAD_A_PLAIN && d->k_split > 0 && d->k_split % 64 == 0 && d->k_split < d->K && d->lda2 % 8 == 0 && al16(d->a2) &&
                       d->a_row_mod >= 0 && d->a2_row_mod >= 0 && d->M < (1LL << 31) && !d->rowstat_in,
                   "apad_gemm: two-source A needs a plain A operand, 0 < k_split < K, k_split %% 64 == 0, lda2 %% 8 == 0");
    else
        APAD_CHECK(d->a_row_mod == 0 || d->a_mode == APAD_A_PLAIN, "apad_gemm: a_row_mod needs a plain A operand");
    p.rs_out = d->rowstat_out; p.rs_in = d->rowstat_in; p.ln_cs = d->ln_colsum; p.ln_bb = d->ln_bias;
    p.rs_in_tiles = d->rowstat_in_tiles; p.rs_out_tiles = (int32_t)((d->N + 63) / 64); p.ln_eps = d->ln_eps;
    if (d->rowstat_out)
        APAD_CHECK(d->out_mode == APAD_OUT_ROWMAJOR && (d->epilogue == APAD_EPI_NONE || d->epilogue == APAD_EPI_SILU || d->epilogue == APAD_EPI_GELU) &&
                       d->N % 64 == 0, "apad_gemm: rowstat_out needs a row-major output with N %% 64 == 0 and no GEGLU");
    if (d->rowstat_in)
        APAD_CHECK(d->ln_colsum && d->ln_bias && d->rowstat_in_tiles > 0 && d->a_mode == APAD_A_PLAIN && !d->bias,
                   "apad_gemm: rowstat_in (folded LayerNorm) needs ln_colsum, ln_bias (which carries the layer's bias), rowstat_in_tiles, a plain A operand");
    hipStream_t s = (hipStream_t)stream;
    {  // 3x3 convolutions whose packed weight form came along: the halo-resident kernel (every row count of an eligible layer)
        const int rc = apad_hconv_try(d, s);
        if (rc <= 0) return rc;
    }
    {  // the compute-bound launches (large-M 3x3 convolutions, plain GEMMs with N % 128 == 0): the big-tile LDS-DMA kernel
        const int rc = apad_cgemm_try(d, s);
        if (rc <= 0) return rc;
    }
    return d->dtype == APAD_BF16 ? dispatch_amode<APAD_BF16>(p, d, s) : dispatch_amode<APAD_F16>(p, d, s);
}
