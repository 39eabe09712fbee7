// The fp32 precision mode of the path (dtype APAD_F32): every contraction on the exact-f32 matrix instruction
// v_mfma_f32_32x32x2_f32 (f32 operands, f32 accumulate, bitwise an fmaf chain -- MI355X_MICROARCH.md "Matrix cores"),
// precise libm exp / erf in the epilogues, no storage rounding anywhere.  This is the mode the reference itself runs
// its AudioMAE encoder (pipeline/pipeline_audioldm2.py:926, never cast), its CPU configuration and its default
// training in, and the mode in which the kernels' arithmetic can be compared with the fp32 oracle chain to rounding.
// It is the ACCURACY mode: the tiles are small and simple (64x64x32 GEMM tiles, waves that stream K / V^T straight
// from L2) -- 157 TFLOP/s is the ceiling of this instruction anyway, 1/16 of the bf16 rate.
//
// Fragment layout of v_mfma_f32_32x32x2_f32 (wave64): A operand lane l = A[l % 32][l / 32], B operand lane l =
// B[l / 32][l % 32], C/D lane l, register r = C[8 * (r / 4) + 4 * (l / 32) + r % 4][l % 32] -- the C layout of every
// 32x32 MFMA, so the transposed-score trick of attention.hip (C layout re-used as the next B operand) carries over:
// register r of the score tile is the B element of k-step r when V^T is read with the same key permutation.
#include <math.h>
#include "common.h"
#include "f32_ops.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float silu_p(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_p(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// "gelu_new" of GPT-2 / T5's gated-gelu (transformers NewGELUActivation)
__device__ __forceinline__ float gelu_tanh_p(float x) { return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x))); }
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------------------
// GEMM: out = epilogue(A . W^T + bias + rowgroup_bias) + residual, all A modes / epilogues / output modes of apad_gemm
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TB = 64;        // block tile edge (4 waves, one 32x32 MFMA tile each)
constexpr int BKF = 32;       // floats per k-tile = one 128-byte LDS row
constexpr int CLD = TB + 4;   // epilogue tile row stride (floats)

struct G32P {
    const float* a;
    const float* w;
    float* out;
    float* out2;
    float* out3;
    const float* bias;
    const float* residual;
    const float* rg;
    const int32_t* step_ptr;
    int64_t M, N, K, lda, ldw, ldo, ldr, ld_rg, rows_per_group;
    int32_t Hin, Win, Cin, Hout, Wout, stride, Hup, Wup, src_batch_mod, res_mod;
    int32_t heads, head_dim, L, Lpad;
    int32_t epi, outmode, n_tiles;
    int32_t taps, dilation, pad, transposed, pre_act;  // APAD_A_CONV1D
    float pre_slope;
    int32_t lead;  // conv3x3: zero rows / columns before the first source row / column (1, or 0 with conv_asym_pad)
};

// float offset of 16-byte chunk `chunk` (0..7) of tile row `row` (the swizzle of gemm.hip's lds_off)
__device__ __forceinline__ int lds32(int row, int chunk) { return row * BKF + ((chunk ^ ((row >> 1) & 7)) << 2); }

struct Row32 {
    int64_t base;  // PLAIN: element offset of the row; CONV / PATCH: source batch index
    int oy, ox;
    bool valid;
};

template <int AMODE> __device__ __forceinline__ f4 load_a32(const G32P& p, const Row32& r, int k) {
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    if (!r.valid || k >= p.K) return z;
    if (AMODE == APAD_A_PLAIN) {
        return *reinterpret_cast<const f4*>(p.a + r.base + k);
    } else if (AMODE == APAD_A_CONV3X3) {
        const int tap = k / p.Cin, c = k - tap * p.Cin;
        const int ky = tap / 3, kx = tap - ky * 3;
        int iy = r.oy * p.stride + ky - p.lead, ix = r.ox * p.stride + kx - p.lead;
        const int H = p.Hup > 0 ? p.Hup : p.Hin, W = p.Hup > 0 ? p.Wup : p.Win;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) return z;
        if (p.Hup > 0) {  // nearest-neighbour source index, floor(dst * in / out)
            iy = (int)(((int64_t)iy * p.Hin) / p.Hup);
            ix = (int)(((int64_t)ix * p.Win) / p.Wup);
        }
        return *reinterpret_cast<const f4*>(p.a + ((r.base * p.Hin + iy) * p.Win + ix) * p.Cin + c);
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
        f4 v = *reinterpret_cast<const f4*>(p.a + ((int64_t)r.base * p.Hin + ti) * p.Cin + c);
        if (p.pre_act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.pre_slope;
        }
        return v;
    } else {  // PATCH16: mel [B][Hin][Win]; k = py * 16 + px
        const int py = k >> 4, px = k & 15;
        return *reinterpret_cast<const f4*>(p.a + (r.base * p.Hin + r.oy * 16 + py) * p.Win + r.ox * 16 + px);
    }
}

template <int AMODE> __global__ __launch_bounds__(256) void gemm_f32_kernel(G32P p) {
    __shared__ __attribute__((aligned(16))) float smem[TB * CLD];  // staging: 2 x 64 x 32 floats; epilogue: 64 x 68
    float* const sA = smem;
    float* const sB = smem + TB * BKF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const bool geglu = p.epi == APAD_EPI_GEGLU || p.epi == APAD_EPI_GEGLU_TANH;
    const int bn_out = geglu ? TB / 2 : TB;  // GEGLU: first half of the tile columns = value rows, second half = gate rows
    const int nt = blockIdx.x % p.n_tiles, mt = blockIdx.x / p.n_tiles;
    const int64_t m0 = (int64_t)mt * TB, n0 = (int64_t)nt * bn_out;
    auto wrow = [&](int nl) -> int64_t {
        if (geglu) return nl < TB / 2 ? n0 + nl : p.N + n0 + (nl - TB / 2);
        return n0 + nl;
    };
    auto wvalid = [&](int nl) -> bool {
        if (geglu) return (nl < TB / 2 ? n0 + nl : n0 + nl - TB / 2) < p.N;
        return n0 + nl < p.N;
    };
    const int chunk = tid & 7;
    Row32 ra[2];
    int64_t wb[2];
    bool wv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = (tid >> 3) + 32 * i;
        const int64_t m = m0 + rl;
        ra[i].valid = m < p.M;
        ra[i].oy = ra[i].ox = 0;
        ra[i].base = 0;
        if (ra[i].valid) {
            if (AMODE == APAD_A_PLAIN) {
                ra[i].base = m * p.lda;
            } else if (AMODE == APAD_A_CONV3X3) {
                const int64_t hw = (int64_t)p.Hout * p.Wout;
                const int64_t b = m / hw;
                const int rem = (int)(m - b * hw);
                ra[i].oy = rem / p.Wout;
                ra[i].ox = rem - ra[i].oy * p.Wout;
                ra[i].base = p.src_batch_mod > 0 ? b % p.src_batch_mod : b;
            } else if (AMODE == APAD_A_CONV1D) {
                const int64_t b = m / p.Hout;
                ra[i].oy = (int)(m - b * p.Hout);
                ra[i].base = b;
            } else {
                const int wp = p.Win >> 4, hp = p.Hin >> 4;
                const int64_t b = m / (hp * wp);
                const int rem = (int)(m - b * hp * wp);
                ra[i].oy = rem / wp;
                ra[i].ox = rem - ra[i].oy * wp;
                ra[i].base = b;
            }
        }
        wv[i] = wvalid(rl);
        wb[i] = wv[i] ? wrow(rl) * p.ldw : 0;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk = (int)((p.K + BKF - 1) / BKF);
    f4 ga[2], gb[2];
    auto gload = [&](int kt) {
        const int k = kt * BKF + chunk * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ga[i] = load_a32<AMODE>(p, ra[i], k);
            const f4 z = {0.f, 0.f, 0.f, 0.f};
            gb[i] = (wv[i] && k < p.K) ? *reinterpret_cast<const f4*>(p.w + wb[i] + k) : z;
        }
    };
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rl = (tid >> 3) + 32 * i;
            *reinterpret_cast<f4*>(sA + lds32(rl, chunk)) = ga[i];
            *reinterpret_cast<f4*>(sB + lds32(rl, chunk)) = gb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);  // in flight under the MFMAs of this k-tile
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ch = ks * 2 + half;
            const f4 af = *reinterpret_cast<const f4*>(sA + lds32(wm * 32 + l31, ch));
            const f4 bf = *reinterpret_cast<const f4*>(sB + lds32(wn * 32 + l31, ch));
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = mfma2(af[j], bf[j], acc);  // k = kt*32 + ch*4 + j in both operands
        }
        __syncthreads();
    }

    // ---- epilogue: acc (+ bias, + rowgroup bias, activation) -> LDS tile ----
    float* const ct = smem;
    const int64_t step = p.step_ptr ? (int64_t)*p.step_ptr : 0;
    const bool one_group = p.rows_per_group >= p.M;
    {
        const int nl = wn * 32 + l31;
        const bool nvalid = wvalid(nl);
        const int64_t wr = nvalid ? wrow(nl) : 0;
        const float bv = (p.bias && nvalid) ? p.bias[wr] : 0.f;
        const float rg0 = (p.rg && one_group && nvalid) ? p.rg[step * p.ld_rg + wr] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = acc[r] + bv + rg0;
            if (p.rg && !one_group) {
                const int64_t m = m0 + ml;
                if (m < p.M && nvalid) v += p.rg[(m / p.rows_per_group + step) * p.ld_rg + wr];
            }
            if (p.epi == APAD_EPI_SILU) v = silu_p(v);
            if (p.epi == APAD_EPI_GELU) v = gelu_p(v);
            if (p.epi == APAD_EPI_TANH) v = tanhf(v);
            if (p.epi == APAD_EPI_RELU) v = fmaxf(v, 0.f);
            if (p.epi == APAD_EPI_GELU_TANH) v = gelu_tanh_p(v);
            ct[ml * CLD + nl] = v;
        }
    }
    __syncthreads();

    const int Cq = (int)(p.N / 3);  // fused q|k|v: the tile lies in exactly one third of the columns (C % 64 == 0)
    const int qseg = (p.outmode == APAD_OUT_QKV) ? (int)(n0 / Cq) : 0;
    if (p.outmode == APAD_OUT_ROWMAJOR || (p.outmode == APAD_OUT_QKV && qseg < 2)) {
        float* const obase = (p.outmode == APAD_OUT_QKV && qseg == 1) ? p.out2 : p.out;
        const int64_t ncol0 = (p.outmode == APAD_OUT_QKV) ? (int64_t)qseg * Cq : 0;
        const int vpr = bn_out / 4;  // 16-byte vectors per output row
        for (int idx = tid; idx < TB * vpr; idx += 256) {
            const int rl = idx / vpr, vc = idx - rl * vpr;
            const int64_t m = m0 + rl, n = n0 + vc * 4;
            if (m >= p.M || n >= p.N) continue;
            f4 f = *reinterpret_cast<const f4*>(&ct[rl * CLD + vc * 4]);
            if (geglu) {
                const f4 g = *reinterpret_cast<const f4*>(&ct[rl * CLD + TB / 2 + vc * 4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) f[e] *= p.epi == APAD_EPI_GEGLU_TANH ? gelu_tanh_p(g[e]) : gelu_p(g[e]);
            }
            if (p.residual) {
                const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
                f += *reinterpret_cast<const f4*>(p.residual + rm * p.ldr + n);
            }
            *reinterpret_cast<f4*>(obase + m * p.ldo + (n - ncol0)) = f;
        }
    } else {  // APAD_OUT_VT (or the v third of APAD_OUT_QKV): consecutive lanes -> consecutive tokens of one (head, dd) row
        float* const o = p.outmode == APAD_OUT_QKV ? p.out3 : p.out;
        const int64_t nsub = (p.outmode == APAD_OUT_QKV) ? 2 * (int64_t)Cq : 0;
        for (int idx = tid; idx < TB * TB; idx += 256) {
            const int nl = idx / TB, rl = idx % TB;
            const int64_t m = m0 + rl;
            int64_t n = n0 + nl;
            if (m >= p.M || n >= p.N) continue;
            n -= nsub;
            const int64_t b = m / p.L;
            const int l = (int)(m - b * p.L);
            const int h = (int)(n / p.head_dim), dd = (int)(n - (int64_t)h * p.head_dim);
            o[((b * p.heads + h) * p.head_dim + dd) * p.Lpad + l] = ct[rl * CLD + nl];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention: softmax(Q K^T * scale + bias) V, one or two independently normalised key segments (attention.hip's
// operator).  One wave = 32 queries of one (batch, head); waves are independent; K fragments (A operand, rows = keys)
// and V^T fragments (A operand of the second MFMA, rows = head channels) are read straight from global memory.
// ---------------------------------------------------------------------------------------------------------------------
struct A32P {
    const float* q;
    const float* k;
    const float* vt;
    const float* k2;
    const float* vt2;
    float* out;
    const float* key_bias;
    float* lse;
    int64_t q_sb, q_sn, k_sb, k_sl, vt_sb, k2_sb, k2_sl, vt2_sb, o_sb, o_sn;
    int32_t B, N, H, L, Lpad, L2, Lpad2, kvdiv, kvdiv2;
    float scale_log2, scale2;
};

constexpr float LOG2E_F = 1.4426950408889634f;
constexpr float NEG_BIG_F = -1.0e30f;

// un-normalised O^T (o), running max m (scaled log2 domain) and denominator of ONE softmax segment over L keys
template <int D>
__device__ __forceinline__ void segment32(const float* kbase, int64_t k_sl, const float* vbase, int L, int Lpad, const float* bias,
                                          float c, const float (&qf)[D / 8][4], f32x16 (&o)[(D + 31) / 32], float& den, float& m,
                                          int l31, int half) {
    constexpr int DT_TILES = (D + 31) / 32;
    float sum = 0.f;
    m = NEG_BIG_F;
#pragma unroll
    for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    for (int key0 = 0; key0 < L; key0 += 32) {
        // S^T (32 keys x 32 queries) = K . Q^T
        const int krow = key0 + l31 < L ? key0 + l31 : L - 1;  // rows past L are masked below
        const float* kp = kbase + (int64_t)krow * k_sl + half * 4;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int cc = 0; cc < D / 8; ++cc) {
            const f4 kf = *reinterpret_cast<const f4*>(kp + cc * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) s = mfma2(kf[j], qf[cc][j], s);
        }
        float tmax = NEG_BIG_F;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = s[r] * c;
            if (bias) v += bias[key < L ? key : L - 1] * LOG2E_F;
            v = key < L ? v : NEG_BIG_F;
            s[r] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = half_max(tmax);
        const float mnew = fmaxf(m, tmax);
        const float alpha = exp2f(m - mnew);
        m = mnew;
        sum *= alpha;
#pragma unroll
        for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = exp2f(s[r] - m);
            sum += s[r];
        }
        // O^T += V^T . P^T : register r of the score tile is the B element of k-step r
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int dt = 0; dt < DT_TILES; ++dt) {
                const int d = dt * 32 + l31;
                f4 vf = {0.f, 0.f, 0.f, 0.f};
                if (d < D) vf = *reinterpret_cast<const f4*>(vbase + (int64_t)d * Lpad + key0 + 8 * g + 4 * half);
#pragma unroll
                for (int j = 0; j < 4; ++j) o[dt] = mfma2(vf[j], s[g * 4 + j], o[dt]);
            }
        }
    }
    den = half_sum(sum);
}

template <int D> __global__ __launch_bounds__(256) void attn_f32_kernel(A32P p) {
    constexpr int DT_TILES = (D + 31) / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int bh = blockIdx.y, h = bh % p.H, b = bh / p.H;
    const int q0 = blockIdx.x * 128 + wave * 32;
    if (q0 >= p.N) return;  // waves are independent: no barrier below
    int qi = q0 + l31;
    const bool qvalid = qi < p.N;
    qi = qvalid ? qi : p.N - 1;
    float qf[D / 8][4];  // Q^T B-operand fragments: lane holds Q[qi][cc*8 + half*4 .. +4)
    const float* qp = p.q + (int64_t)b * p.q_sb + (int64_t)qi * p.q_sn + h * D + half * 4;
#pragma unroll
    for (int cc = 0; cc < D / 8; ++cc) {
        const f4 v = *reinterpret_cast<const f4*>(qp + cc * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) qf[cc][j] = v[j];
    }
    f32x16 o[DT_TILES];
    float den, m;
    {
        const int bk = b / p.kvdiv;
        const float* bias = p.key_bias ? p.key_bias + (int64_t)b * p.L : nullptr;
        segment32<D>(p.k + (int64_t)bk * p.k_sb + h * D, p.k_sl, p.vt + (int64_t)bk * p.vt_sb + (int64_t)h * D * p.Lpad, p.L, p.Lpad,
                     bias, p.scale_log2, qf, o, den, m, l31, half);
    }
    if (p.lse != nullptr && half == 0 && q0 + l31 < ((p.N + 31) & ~31))  // (pad entries: 0, see attention.hip)
        p.lse[((int64_t)b * p.H + h) * ((p.N + 31) & ~31) + q0 + l31] = qvalid ? m + log2f(den) : 0.f;
    const float inv = 1.0f / den;
#pragma unroll
    for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= inv;
    if (p.L2 > 0) {
        f32x16 o2[DT_TILES];
        float den2, m2;
        const int bk = b / p.kvdiv2;
        segment32<D>(p.k2 + (int64_t)bk * p.k2_sb + h * D, p.k2_sl, p.vt2 + (int64_t)bk * p.vt2_sb + (int64_t)h * D * p.Lpad2, p.L2,
                     p.Lpad2, nullptr, p.scale_log2, qf, o2, den2, m2, l31, half);
        const float inv2 = 1.0f / den2;
#pragma unroll
        for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] += p.scale2 * (o2[dt][r] * inv2);  // attention_processor.py:454
    }
    if (!qvalid) return;
    float* ob = p.out + (int64_t)b * p.o_sb + (int64_t)qi * p.o_sn + h * D;
#pragma unroll
    for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = dt * 32 + 8 * g + 4 * half;
            if (d0 < D) {
                const f4 v = {o[dt][g * 4], o[dt][g * 4 + 1], o[dt][g * 4 + 2], o[dt][g * 4 + 3]};
                *reinterpret_cast<f4*>(ob + d0) = v;
            }
        }
}

template <int D> int attn_launch(const A32P& p, hipStream_t s) {
    dim3 grid((unsigned)((p.N + 127) / 128), (unsigned)(p.B * p.H));
    hipLaunchKernelGGL((attn_f32_kernel<D>), grid, dim3(256), 0, s, p);
    return apad_check_launch("apad_attention(f32)");
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm (one wave per row) and GroupNorm over NHWC (one workgroup per (group, sample)): mean first, then the sum of
// squared deviations -- the two-pass form, which is what an fp32 comparison to rounding wants; fixed-order reductions
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_f32_kernel(const float* x, const float* gamma, const float* beta, float* out,
                                                            int64_t M, int C, int64_t ldx, int64_t ldo, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ldx;
    float sum = 0.f;
    for (int i = lane * 4; i < C; i += 256) {
        const f4 v = *reinterpret_cast<const f4*>(xr + i);
        sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
    for (int i = lane * 4; i < C; i += 256) {
        const f4 v = *reinterpret_cast<const f4*>(xr + i) - mean;
        sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + eps);
    for (int i = lane * 4; i < C; i += 256) {
        const f4 v = *reinterpret_cast<const f4*>(xr + i);
        const f4 g = *reinterpret_cast<const f4*>(gamma + i), b = *reinterpret_cast<const f4*>(beta + i);
        *reinterpret_cast<f4*>(out + row * ldo + i) = (v - mean) * rstd * g + b;
    }
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();  // red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void groupnorm_f32_kernel(const float* x, const float* gamma, const float* beta, float* out, int HW,
                                                            int C, int G, float eps, int silu) {
    __shared__ float red[4];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int cg = C / G, vpg = cg >> 2;  // 16-byte vectors of one pixel's slice of this group
    const int nvec = HW * vpg;
    const float* xb = x + (int64_t)b * HW * C + g * cg;
    float* ob = out + (int64_t)b * HW * C + g * cg;
    float sum = 0.f;
    for (int idx = tid; idx < nvec; idx += 256) {
        const int px = idx / vpg, vc = idx - px * vpg;
        const f4 v = *reinterpret_cast<const f4*>(xb + (int64_t)px * C + vc * 4);
        sum += (v[0] + v[1]) + (v[2] + v[3]);
    }
    const float n = (float)HW * (float)cg;
    const float mean = block_sum256(sum, red) / n;
    float sq = 0.f;
    for (int idx = tid; idx < nvec; idx += 256) {
        const int px = idx / vpg, vc = idx - px * vpg;
        const f4 v = *reinterpret_cast<const f4*>(xb + (int64_t)px * C + vc * 4) - mean;
        sq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float rstd = 1.0f / sqrtf(block_sum256(sq, red) / n + eps);
    for (int idx = tid; idx < nvec; idx += 256) {
        const int px = idx / vpg, vc = idx - px * vpg;
        const f4 v = *reinterpret_cast<const f4*>(xb + (int64_t)px * C + vc * 4);
        const f4 gm = *reinterpret_cast<const f4*>(gamma + g * cg + vc * 4), bt = *reinterpret_cast<const f4*>(beta + g * cg + vc * 4);
        f4 y = (v - mean) * rstd * gm + bt;
        if (silu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = silu_p(y[e]);
        }
        *reinterpret_cast<f4*>(ob + (int64_t)px * C + vc * 4) = y;
    }
}

// rep [B][513][768] -> out [B][(64/tp)*(8/fp)][768], (avg + max) / 2 over (tp x fp) windows (AudioMAE.py:148-182)
__global__ __launch_bounds__(256) void pool_f32_kernel(const float* rep, float* out, int B, int tp, int fp) {
    const int nt = 64 / tp, nf = 8 / fp, La = nt * nf;
    const int64_t total = (int64_t)B * La * 192;  // 192 vectors of 4 channels
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int vc = (int)(idx % 192);
        const int64_t tok = idx / 192;
        const int b = (int)(tok / La), o = (int)(tok % La);
        const int ot = o / nf, of = o % nf;
        f4 s = {0.f, 0.f, 0.f, 0.f}, mx = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        for (int dt = 0; dt < tp; ++dt)
            for (int df = 0; df < fp; ++df) {
                const int row = 1 + 8 * (ot * tp + dt) + (of * fp + df);
                const f4 v = *reinterpret_cast<const f4*>(rep + ((int64_t)b * 513 + row) * 768 + vc * 4);
                s += v;
#pragma unroll
                for (int e = 0; e < 4; ++e) mx[e] = fmaxf(mx[e], v[e]);
            }
        *reinterpret_cast<f4*>(out + tok * 768 + vc * 4) = (s / (float)(tp * fp) + mx) / 2.0f;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// fp32 attention backward (the reference's default training precision: train.sh leaves --mixed_precision unset).  Plain fp32
// FMAs, no MFMA: P is rebuilt from q, k and the forward's log-sum-exp; one wave owns one query row (dq) or one key row (dk, dv),
// its lanes walk the other sequence, head-dim accumulators are reduced across the wave at the end.  The head dimension is
// processed in chunks of 32 accumulators per lane (scores are recomputed per chunk).  A precision mode: ~3 N L D scalar MACs per
// (sample, head), two launches per attention.
struct BwdF {
    const float *q, *k, *v, *out, *dout, *lse, *key_bias;
    float *delta, *dq, *dk, *dv;
    int B, N, H, L, D, Npad;
    float scale, scale_log2, dout_scale;
    int accumulate_dq;
};
// (LOG2E_F: defined above)

__global__ __launch_bounds__(256) void delta_f32_kernel(BwdF p) {  // delta[b][h][n] = dout_scale * sum_d dO . O
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.B * p.N * p.H) return;
    const int h = (int)(idx % p.H);
    const int64_t bn = idx / p.H;
    const int n = (int)(bn % p.N), b = (int)(bn / p.N);
    const int C = p.H * p.D;
    const float* o = p.out + bn * C + h * p.D;
    const float* g = p.dout + bn * C + h * p.D;
    float acc = 0.f;
    for (int d = 0; d < p.D; ++d) acc = fmaf(o[d], g[d], acc);
    p.delta[((int64_t)b * p.H + h) * p.Npad + n] = acc * p.dout_scale;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// dq[b][i][h] = scale * sum_j P_ij (dP_ij - delta_i) k_j ; wave = one (b, h, i)
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_kernel(BwdF p) {
    __shared__ float sh[4][2 * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // (b, h, i)
    if (row >= (int64_t)p.B * p.H * p.N) return;
    const int i = (int)(row % p.N);
    const int64_t bh = row / p.N;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const int C = p.H * p.D, D = p.D;
    float* qs = sh[wave];
    float* gs = sh[wave] + 128;
    for (int d = lane; d < D; d += 64) {
        qs[d] = p.q[((int64_t)b * p.N + i) * C + h * D + d];
        gs[d] = p.dout[((int64_t)b * p.N + i) * C + h * D + d];
    }
    __builtin_amdgcn_wave_barrier();
    const float lse = p.lse[bh * p.Npad + i], delta = p.delta[bh * p.Npad + i];
    const float* kb = p.k + (int64_t)b * p.L * C + h * D;
    const float* vb = p.v + (int64_t)b * p.L * C + h * D;
    const float* bias = p.key_bias ? p.key_bias + (int64_t)b * p.L : nullptr;
    float* dst = p.dq + ((int64_t)b * p.N + i) * C + h * D;
    for (int c0 = 0; c0 < D; c0 += 32) {
        float acc[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) acc[t] = 0.f;
        for (int j = lane; j < p.L; j += 64) {
            const float* kr = kb + (int64_t)j * C;
            const float* vr = vb + (int64_t)j * C;
            float sc = 0.f, dp = 0.f;
            for (int d = 0; d < D; ++d) {
                sc = fmaf(qs[d], kr[d], sc);
                dp = fmaf(gs[d], vr[d], dp);
            }
            const float pij = exp2f(sc * p.scale_log2 + (bias ? bias[j] * LOG2E_F : 0.f) - lse);
            const float ds = pij * (dp * p.dout_scale - delta) * p.scale;
#pragma unroll
            for (int t = 0; t < 32; ++t)
                if (c0 + t < D) acc[t] = fmaf(ds, kr[c0 + t], acc[t]);
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const float v = wave_sum_f(acc[t]);
            if (lane == 0 && c0 + t < D) dst[c0 + t] = p.accumulate_dq ? dst[c0 + t] + v : v;
        }
    }
}

// dk[b][j][h] = scale * sum_i P_ij (dP_ij - delta_i) q_i ; dv[b][j][h] = dout_scale * sum_i P_ij dO_i ; wave = one (b, h, j)
__global__ __launch_bounds__(256) void attn_bwd_dkv_f32_kernel(BwdF p) {
    __shared__ float sh[4][2 * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = (int64_t)blockIdx.x * 4 + wave;  // (b, h, j)
    if (row >= (int64_t)p.B * p.H * p.L) return;
    const int j = (int)(row % p.L);
    const int64_t bh = row / p.L;
    const int h = (int)(bh % p.H), b = (int)(bh / p.H);
    const int C = p.H * p.D, D = p.D;
    float* ks = sh[wave];
    float* vs = sh[wave] + 128;
    for (int d = lane; d < D; d += 64) {
        ks[d] = p.k[((int64_t)b * p.L + j) * C + h * D + d];
        vs[d] = p.v[((int64_t)b * p.L + j) * C + h * D + d];
    }
    __builtin_amdgcn_wave_barrier();
    const float bj = p.key_bias ? p.key_bias[(int64_t)b * p.L + j] * LOG2E_F : 0.f;
    const float* qb = p.q + (int64_t)b * p.N * C + h * D;
    const float* gb = p.dout + (int64_t)b * p.N * C + h * D;
    float* dkd = p.dk + ((int64_t)b * p.L + j) * C + h * D;
    float* dvd = p.dv + ((int64_t)b * p.L + j) * C + h * D;
    for (int c0 = 0; c0 < D; c0 += 32) {
        float ak[32], av[32];
#pragma unroll
        for (int t = 0; t < 32; ++t) ak[t] = av[t] = 0.f;
        for (int i = lane; i < p.N; i += 64) {
            const float* qr = qb + (int64_t)i * C;
            const float* gr = gb + (int64_t)i * C;
            float sc = 0.f, dp = 0.f;
            for (int d = 0; d < D; ++d) {
                sc = fmaf(qr[d], ks[d], sc);
                dp = fmaf(gr[d], vs[d], dp);
            }
            const float pij = exp2f(sc * p.scale_log2 + bj - p.lse[bh * p.Npad + i]);
            const float ds = pij * (dp * p.dout_scale - p.delta[bh * p.Npad + i]) * p.scale;
            const float pv = pij * p.dout_scale;
#pragma unroll
            for (int t = 0; t < 32; ++t)
                if (c0 + t < D) {
                    ak[t] = fmaf(ds, qr[c0 + t], ak[t]);
                    av[t] = fmaf(pv, gr[c0 + t], av[t]);
                }
        }
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const float a = wave_sum_f(ak[t]), c = wave_sum_f(av[t]);
            if (lane == 0 && c0 + t < D) {
                dkd[c0 + t] = a;
                dvd[c0 + t] = c;
            }
        }
    }
}

}  // namespace

int apad_f32_gemm(const apad_gemm_desc* d, hipStream_t s) {
    APAD_CHECK(d->a && d->w && d->out, "apad_gemm(f32): null operand");
    APAD_CHECK(d->M > 0 && d->N > 0 && d->K > 0, "apad_gemm(f32): empty problem M=%lld N=%lld K=%lld", (long long)d->M, (long long)d->N,
               (long long)d->K);
    APAD_CHECK(d->K % 4 == 0 && d->ldw % 4 == 0, "apad_gemm(f32): K and ldw must be multiples of 4 (K=%lld ldw=%lld)", (long long)d->K,
               (long long)d->ldw);
    APAD_CHECK(al16(d->a) && al16(d->w) && al16(d->out) && al16(d->residual), "apad_gemm(f32): pointers must be 16-byte aligned");
    G32P p;
    p.a = (const float*)d->a; p.w = (const float*)d->w; p.out = (float*)d->out; p.out2 = (float*)d->out2; p.out3 = (float*)d->out3;
    p.bias = (const float*)d->bias; p.residual = (const float*)d->residual; p.rg = (const float*)d->rowgroup_bias;
    p.step_ptr = d->step_ptr;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldw = d->ldw; p.ldo = d->ldo; p.ldr = d->ldr; p.ld_rg = d->ld_rg;
    p.rows_per_group = d->rows_per_group > 0 ? d->rows_per_group : 1;
    p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.Hout = d->Hout; p.Wout = d->Wout;
    p.stride = d->stride; p.Hup = d->Hup; p.Wup = d->Wup; p.src_batch_mod = d->src_batch_mod; p.res_mod = d->residual_row_mod;
    p.heads = d->heads; p.head_dim = d->head_dim; p.L = d->L; p.Lpad = d->Lpad;
    p.epi = d->epilogue; p.outmode = d->out_mode;
    APAD_CHECK(d->epilogue >= APAD_EPI_NONE && d->epilogue <= APAD_EPI_GEGLU_TANH, "apad_gemm(f32): unknown epilogue %d", d->epilogue);
    p.taps = d->taps; p.dilation = d->dilation; p.pad = d->pad; p.transposed = d->transposed; p.pre_act = d->a_pre_act;
    p.pre_slope = d->a_pre_slope;
    p.lead = d->conv_asym_pad ? 0 : 1;
    if (d->a_mode == APAD_A_PLAIN) {
        APAD_CHECK(d->lda % 4 == 0, "apad_gemm(f32): lda must be a multiple of 4");
    } else if (d->a_mode == APAD_A_CONV3X3) {
        APAD_CHECK(d->epilogue == APAD_EPI_NONE && d->out_mode == APAD_OUT_ROWMAJOR,
                   "apad_gemm(f32): conv3x3 supports epilogue NONE / row-major output only");
        APAD_CHECK(d->Cin > 0 && d->Cin % 4 == 0 && d->K == 9LL * d->Cin, "apad_gemm(f32): conv3x3 needs Cin%%4==0 and K==9*Cin");
        APAD_CHECK(d->stride == 1 || d->stride == 2, "apad_gemm(f32): conv stride must be 1 or 2");
        APAD_CHECK(d->Hin > 0 && d->Win > 0 && d->Hout > 0 && d->Wout > 0 && d->M % ((int64_t)d->Hout * d->Wout) == 0,
                   "apad_gemm(f32): conv geometry inconsistent with M");
        APAD_CHECK((d->Hup > 0) == (d->Wup > 0), "apad_gemm(f32): Hup/Wup must both be set or both 0");
    } else if (d->a_mode == APAD_A_PATCH16) {
        APAD_CHECK(d->epilogue == APAD_EPI_NONE && d->out_mode == APAD_OUT_ROWMAJOR,
                   "apad_gemm(f32): patch16 supports epilogue NONE / row-major output only");
        APAD_CHECK(d->K == 256 && d->Hin % 16 == 0 && d->Win % 16 == 0, "apad_gemm(f32): patch16 needs K==256 and H,W %% 16 == 0");
        APAD_CHECK(d->M % ((int64_t)(d->Hin / 16) * (d->Win / 16)) == 0, "apad_gemm(f32): patch16 M inconsistent");
    } else if (d->a_mode == APAD_A_CONV1D) {
        APAD_CHECK((d->epilogue == APAD_EPI_NONE || d->epilogue == APAD_EPI_TANH) && d->out_mode == APAD_OUT_ROWMAJOR,
                   "apad_gemm(f32): conv1d supports epilogue NONE / TANH and row-major output only");
        APAD_CHECK(d->Cin > 0 && d->Cin % 4 == 0 && d->taps > 0 && d->K == (int64_t)d->taps * d->Cin, "apad_gemm(f32): conv1d needs Cin%%4==0 and K==taps*Cin");
        APAD_CHECK(d->Hin > 0 && d->Hout > 0 && d->M % d->Hout == 0 && d->pad >= 0, "apad_gemm(f32): conv1d geometry inconsistent with M");
        APAD_CHECK(d->transposed ? d->stride >= 1 : d->dilation >= 1, "apad_gemm(f32): conv1d needs dilation >= 1 (stride >= 1 when transposed)");
    } else {
        apad_set_error("apad_gemm(f32): unknown a_mode %d", d->a_mode);
        return -1;
    }
    if (d->out_mode == APAD_OUT_ROWMAJOR) {
        APAD_CHECK(d->N % 4 == 0 && d->ldo % 4 == 0, "apad_gemm(f32): N and ldo must be multiples of 4");
        if (d->residual) APAD_CHECK(d->ldr % 4 == 0, "apad_gemm(f32): ldr must be a multiple of 4");
        if (d->epilogue == APAD_EPI_GEGLU || d->epilogue == APAD_EPI_GEGLU_TANH) APAD_CHECK(d->N % 32 == 0, "apad_gemm(f32): GEGLU needs N %% 32 == 0");
    } else if (d->out_mode == APAD_OUT_QKV) {
        APAD_CHECK(d->epilogue == APAD_EPI_NONE && d->a_mode == APAD_A_PLAIN, "apad_gemm(f32): APAD_OUT_QKV supports plain A / epilogue NONE only");
        APAD_CHECK(d->out2 && d->out3 && al16(d->out2) && al16(d->out3), "apad_gemm(f32): APAD_OUT_QKV needs 16-byte aligned out2 / out3");
        APAD_CHECK(d->heads > 0 && d->head_dim > 0 && d->L > 0 && d->Lpad >= d->L && d->N == 3LL * d->heads * d->head_dim &&
                       d->M % d->L == 0 && (d->N / 3) % 64 == 0 && d->ldo % 4 == 0,
                   "apad_gemm(f32): fused q|k|v geometry inconsistent (needs C %% 64 == 0)");
        APAD_CHECK(!d->residual, "apad_gemm(f32): fused q|k|v takes no residual");
    } else if (d->out_mode == APAD_OUT_VT) {
        APAD_CHECK(d->epilogue == APAD_EPI_NONE, "apad_gemm(f32): APAD_OUT_VT supports epilogue NONE only");
        APAD_CHECK(d->heads > 0 && d->head_dim > 0 && d->L > 0 && d->Lpad >= d->L && d->N == (int64_t)d->heads * d->head_dim &&
                       d->M % d->L == 0,
                   "apad_gemm(f32): V^T output geometry inconsistent");
        APAD_CHECK(!d->residual, "apad_gemm(f32): V^T output takes no residual");
    } else {
        apad_set_error("apad_gemm(f32): unknown out_mode %d", d->out_mode);
        return -1;
    }
    if (d->rowgroup_bias) APAD_CHECK(d->ld_rg > 0, "apad_gemm(f32): rowgroup_bias needs ld_rg");
    const int bn_out = (d->epilogue == APAD_EPI_GEGLU || d->epilogue == APAD_EPI_GEGLU_TANH) ? TB / 2 : TB;
    p.n_tiles = (int)((d->N + bn_out - 1) / bn_out);
    const int64_t m_tiles = (d->M + TB - 1) / TB;
    dim3 grid((unsigned)(p.n_tiles * m_tiles));
    switch (d->a_mode) {
        case APAD_A_PLAIN: hipLaunchKernelGGL((gemm_f32_kernel<APAD_A_PLAIN>), grid, dim3(256), 0, s, p); break;
        case APAD_A_CONV3X3: hipLaunchKernelGGL((gemm_f32_kernel<APAD_A_CONV3X3>), grid, dim3(256), 0, s, p); break;
        case APAD_A_CONV1D: hipLaunchKernelGGL((gemm_f32_kernel<APAD_A_CONV1D>), grid, dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_f32_kernel<APAD_A_PATCH16>), grid, dim3(256), 0, s, p); break;
    }
    return apad_check_launch("apad_gemm(f32)");
}

int apad_f32_attention(const apad_attn_desc* d, hipStream_t s) {
    APAD_CHECK(d->q && d->k && d->vt && d->out, "apad_attention(f32): null operand");
    APAD_CHECK(d->B > 0 && d->N > 0 && d->H > 0 && d->L > 0, "apad_attention(f32): empty problem B=%d N=%d H=%d L=%d", d->B, d->N, d->H,
               d->L);
    APAD_CHECK(d->Lpad >= d->L && d->Lpad % 32 == 0, "apad_attention(f32): Lpad must be >= L and a multiple of 32");
    APAD_CHECK(d->kv_batch_div >= 1, "apad_attention(f32): kv_batch_div must be >= 1");
    APAD_CHECK(al16(d->q) && al16(d->k) && al16(d->vt) && al16(d->out) && al16(d->k2) && al16(d->vt2),
               "apad_attention(f32): pointers must be 16-byte aligned");
    APAD_CHECK(d->q_stride_n % 4 == 0 && d->q_stride_b % 4 == 0 && d->k_stride_l % 4 == 0 && d->k_stride_b % 4 == 0 &&
                   d->o_stride_n % 4 == 0 && d->o_stride_b % 4 == 0 && d->vt_stride_b % 4 == 0,
               "apad_attention(f32): strides must keep 16-byte alignment");
    const bool dual = d->L2 > 0;
    if (dual) {
        APAD_CHECK(d->k2 && d->vt2, "apad_attention(f32): segment 2 needs k2/vt2");
        APAD_CHECK(d->Lpad2 >= d->L2 && d->Lpad2 % 32 == 0, "apad_attention(f32): Lpad2 must be >= L2 and a multiple of 32");
        APAD_CHECK(d->kv2_batch_div >= 1, "apad_attention(f32): kv2_batch_div must be >= 1");
        APAD_CHECK(d->k2_stride_l % 4 == 0 && d->k2_stride_b % 4 == 0 && d->vt2_stride_b % 4 == 0,
                   "apad_attention(f32): segment-2 strides must keep 16-byte alignment");
    }
    APAD_CHECK(!(dual && d->lse), "apad_attention(f32): lse is only defined for a single softmax segment");
    A32P p;
    p.q = (const float*)d->q; p.k = (const float*)d->k; p.vt = (const float*)d->vt;
    p.k2 = (const float*)d->k2; p.vt2 = (const float*)d->vt2; p.out = (float*)d->out;
    p.key_bias = d->key_bias; p.lse = d->lse;
    p.q_sb = d->q_stride_b; p.q_sn = d->q_stride_n; p.k_sb = d->k_stride_b; p.k_sl = d->k_stride_l; p.vt_sb = d->vt_stride_b;
    p.k2_sb = d->k2_stride_b; p.k2_sl = d->k2_stride_l; p.vt2_sb = d->vt2_stride_b; p.o_sb = d->o_stride_b; p.o_sn = d->o_stride_n;
    p.B = d->B; p.N = d->N; p.H = d->H; p.L = d->L; p.Lpad = d->Lpad; p.L2 = dual ? d->L2 : 0; p.Lpad2 = d->Lpad2;
    p.kvdiv = d->kv_batch_div; p.kvdiv2 = dual ? d->kv2_batch_div : 1;
    p.scale_log2 = d->q_prescaled ? 1.0f : d->softmax_scale * LOG2E_F;
    p.scale2 = d->scale2;
    switch (d->D) {
        case 16: return attn_launch<16>(p, s);
        case 32: return attn_launch<32>(p, s);
        case 48: return attn_launch<48>(p, s);
        case 64: return attn_launch<64>(p, s);
        case 80: return attn_launch<80>(p, s);
        case 96: return attn_launch<96>(p, s);
        case 128: return attn_launch<128>(p, s);
    }
    apad_set_error("apad_attention(f32): head dim %d not supported (16,32,48,64,80,96,128)", d->D);
    return -1;
}

int apad_f32_layernorm(const void* x, const void* gamma, const void* beta, void* out, int64_t M, int32_t C, int64_t ldx, int64_t ldo,
                       float eps, hipStream_t s) {
    APAD_CHECK(M > 0 && C > 0 && C % 4 == 0, "apad_layernorm(f32): need M>0, C%%4==0 (M=%lld C=%d)", (long long)M, C);
    APAD_CHECK(ldx % 4 == 0 && ldo % 4 == 0 && al16(x) && al16(out) && al16(gamma) && al16(beta),
               "apad_layernorm(f32): rows must be 16-byte aligned");
    hipLaunchKernelGGL(layernorm_f32_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, (const float*)x, (const float*)gamma,
                       (const float*)beta, (float*)out, M, C, ldx, ldo, eps);
    return apad_check_launch("apad_layernorm(f32)");
}

int apad_f32_groupnorm(const void* x, const void* gamma, const void* beta, void* out, int32_t B, int32_t HW, int32_t C, int32_t G,
                       float eps, int32_t silu, hipStream_t s) {
    APAD_CHECK(B > 0 && HW > 0 && G > 0 && C % G == 0 && (C / G) % 4 == 0, "apad_groupnorm(f32): need C%%G==0, (C/G)%%4==0 (C=%d G=%d)", C,
               G);
    APAD_CHECK(al16(x) && al16(out) && al16(gamma) && al16(beta), "apad_groupnorm(f32): pointers must be 16-byte aligned");
    hipLaunchKernelGGL(groupnorm_f32_kernel, dim3((unsigned)G, (unsigned)B), dim3(256), 0, s, (const float*)x, (const float*)gamma,
                       (const float*)beta, (float*)out, HW, C, G, eps, silu);
    return apad_check_launch("apad_groupnorm(f32)");
}

int apad_f32_audiomae_pool(const void* rep, void* out, int32_t B, int32_t tp, int32_t fp, hipStream_t s) {
    const int64_t total = (int64_t)B * (64 / tp) * (8 / fp) * 192;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pool_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)rep, (float*)out, B, tp, fp);
    return apad_check_launch("apad_audiomae_pool(f32)");
}

int apad_f32_attention_bwd(const apad_attn_bwd_desc* d, hipStream_t s) {
    APAD_CHECK(d->q && d->k && d->v && d->out && d->dout && d->lse && d->delta && d->dq, "apad_attention_bwd(f32): null operand");
    APAD_CHECK(d->B > 0 && d->N > 0 && d->H > 0 && d->L > 0 && d->D > 0 && d->D <= 128, "apad_attention_bwd(f32): empty problem / head dim > 128");
    APAD_CHECK(d->Npad >= d->N, "apad_attention_bwd(f32): Npad must cover N");
    APAD_CHECK((d->dk == nullptr) == (d->dv == nullptr), "apad_attention_bwd(f32): dk and dv are requested together");
    BwdF p;
    p.q = (const float*)d->q; p.k = (const float*)d->k; p.v = (const float*)d->v; p.out = (const float*)d->out;
    p.dout = (const float*)d->dout; p.lse = d->lse; p.key_bias = d->key_bias; p.delta = d->delta;
    p.dq = (float*)d->dq; p.dk = (float*)d->dk; p.dv = (float*)d->dv;
    p.B = d->B; p.N = d->N; p.H = d->H; p.L = d->L; p.D = d->D; p.Npad = d->Npad;
    p.scale = d->softmax_scale; p.scale_log2 = d->softmax_scale * LOG2E_F; p.dout_scale = d->dout_scale;
    p.accumulate_dq = d->accumulate_dq;
    const int64_t nd = (int64_t)p.B * p.N * p.H;
    hipLaunchKernelGGL(delta_f32_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(attn_bwd_dq_f32_kernel, dim3((unsigned)((nd + 3) / 4)), dim3(256), 0, s, p);
    if (p.dk) {
        const int64_t nk = (int64_t)p.B * p.L * p.H;
        hipLaunchKernelGGL(attn_bwd_dkv_f32_kernel, dim3((unsigned)((nk + 3) / 4)), dim3(256), 0, s, p);
    }
    return apad_check_launch("apad_attention_bwd(f32)");
}
