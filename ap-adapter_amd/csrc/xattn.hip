// apad_fused_cross_attention: a whole cross-attention sub-layer of a BasicTransformerBlock in ONE kernel, for key / value
// sets that were hoisted out of the denoise loop and are short (<= 64 keys per segment) -- the adapter's decoupled
// cross-attention (IPAttnProcessor2_0, attention_processor.py:347-470: 8 text keys + La <= 64 audio keys blended by
// ap_scale) and the 16-token T5 cross-attention (AttnProcessor2_0, :214-294):
//     out = x + to_out( A(q, K1, V1, bias) [+ scale2 * A(q, K2, V2)] ) + b_out,   q = to_q(LayerNorm(x))
// SURVEY 8d's "fused q-proj + attn + blend + out-proj": per sample-forward the launch reads x and writes out once
// (2 x N x C x 2 bytes) instead of six activation passes through three kernels.
//
// One wave owns 32 tokens of ONE sample.  The LayerNorm-ed x panel stays in registers as the B operand; per head the
// chain  q_h^T = Wq_h . x^T  ->  S^T = K . q_h  ->  softmax  ->  O_h^T = V^T . P^T  ->  out^T += Wo[:, h] . O_h^T
// never leaves registers: each MFMA's C layout (lane = token, registers = rows (r&3) + 8(r>>2) + 4*half) is the next
// MFMA's B operand, the other operand being read with the matching permuted k order (two 8-byte pieces per lane) -- the
// register trick of apad_attention and apad_geglu_mlp applied three times in a row.  The Wq rows of head h and the
// Wo columns of head h are staged through double-buffered LDS by the whole workgroup, one head ahead; K and V^T
// fragments (a few KB per sample, L2-resident) are read straight from global memory.
// Envelope: C = 256, 8 heads (d = 32): the 1000-token level, where the cross-attention sub-layers cost most.
#include "rp_shared.h"

namespace {

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr float XA_LOG2E = 1.4426950408889634f;
constexpr float XA_NEG_BIG = -1.0e30f;

constexpr int XC = 256, XKC = 16, XH = 8, XD = 32;
constexpr int WQ_ROWB = Cfg<XKC>::ROWB;      // 528: Wq tile row stride
constexpr int WQ_BYTES = 32 * WQ_ROWB;       // 32 rows of Wq (one head)
constexpr int WO_ROWB = 72;                  // Wo head slice: 32 columns (64 B) + 8 B pad: conflict-free 8-byte reads
constexpr int WO_BYTES = XC * WO_ROWB;
constexpr int XSTAGE = WQ_BYTES + WO_BYTES;  // 35 328 B per head

struct XaP {
    const uint8_t* x;
    const uint8_t* gamma;
    const uint8_t* beta;
    const uint8_t* wq;
    const uint8_t* wo;
    const uint8_t* bo;
    const uint8_t* k1;
    const uint8_t* v1t;
    const float* bias1;
    const uint8_t* k2;
    const uint8_t* v2t;
    uint8_t* out;
    int32_t B, N, L1, Lpad1, L2, Lpad2;
    float eps, scale_log2, scale2;
};

// K / V^T fragments of one short segment for one head, straight from global memory (L2-resident).  Issued at the top of a
// head's iteration so that their latency hides under the q-projection MFMAs: with one wave per SIMD nothing else would.
template <int DT> struct XaFrags {
    typename ET<DT>::v8 kf[2][2];  // [32-key sub-tile][K = 16 step over the head dim]
    typename ET<DT>::v8 vf[4];     // [K = 16 step over the keys]
};

template <int DT>
__device__ __forceinline__ void xa_prefetch(XaFrags<DT>& f, const uint8_t* kbase /* K[b] + h*D, row stride XC */,
                                            const uint8_t* vbase /* V^T[b][h] */, int L, int Lpad, int l31, int half) {
    const int nsub = L > 32 ? 2 : 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nsub) break;
        const int key = u * 32 + l31;
        const uint8_t* kp = kbase + ((int64_t)(key < L ? key : L - 1) * XC + 4 * half) * 2;  // rows past L: masked later
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const uint2 lo = *reinterpret_cast<const uint2*>(kp + kk * 32);
            const uint2 hi = *reinterpret_cast<const uint2*>(kp + kk * 32 + 16);
            f.kf[u][kk] = as_v8<DT>(make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
    }
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        if (st >= 2 * nsub) break;
        const uint8_t* vp = vbase + ((int64_t)l31 * Lpad + st * 16 + 4 * half) * 2;  // row = head dim l31 (D = 32: one tile)
        const uint2 v0 = *reinterpret_cast<const uint2*>(vp);
        const uint2 v1 = *reinterpret_cast<const uint2*>(vp + 16);
        f.vf[st] = as_v8<DT>(make_uint4(v0.x, v0.y, v1.x, v1.y));
    }
}

// one short softmax segment for the current head: scores from qb (B operand, k = head dim in C-layout order), result
// O^T (un-normalised) accumulated into o, inv_den = 1 / row sum
template <int DT>
__device__ __forceinline__ void xa_segment(const XaFrags<DT>& f, int L, const float* bias, float c, const typename ET<DT>::v8 (&qb)[2],
                                           f32x16& o, float& inv_den, int half) {
    using E = ET<DT>;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nsub = L > 32 ? 2 : 1;
    f32x16 s[2];
    s[0] = s[1] = zero16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nsub) break;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) s[u] = E::mfma32(f.kf[u][kk], qb[kk], s[u]);
    }
    float tmax = XA_NEG_BIG;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (u >= nsub) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = s[u][r] * c;
            if (bias) v += bias[key < L ? key : L - 1] * XA_LOG2E;
            v = key < L ? v : XA_NEG_BIG;
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
            const float e = __builtin_amdgcn_exp2f(s[u][r] - tmax);
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
        o = E::mfma32(f.vf[st], pf, o);
    }
}

template <int DT, bool DUAL>
__global__ __launch_bounds__(256, 1) void xattn_kernel(XaP p) {
    using E = ET<DT>;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // panels never straddle samples: sample b owns ceil(N / 32) panels, 4 consecutive panels per workgroup
    const int ppn = (p.N + 31) >> 5;
    const int64_t panel = (int64_t)blockIdx.x * 4 + wave;
    const int b = (int)(panel / ppn);
    const int q0 = (int)(panel - (int64_t)b * ppn) * 32;
    const bool active = b < p.B;  // tail workgroup: inactive waves still take part in staging and barriers
    const int bb = active ? b : p.B - 1;
    const int64_t row0 = (int64_t)bb * p.N + q0;    // first global row of the panel
    const int64_t rowend = (int64_t)bb * p.N + p.N;  // rows of this sample end here

    uint8_t* const scr = smem + 2 * XSTAGE + wave * SCR_BYTES;
    float* const lbo = reinterpret_cast<float*>(smem + 2 * XSTAGE + 4 * SCR_BYTES);  // [C] output bias

    // ---- staging (global -> registers -> LDS), one head ahead: Wq rows h*32.., Wo columns h*32.. ----
    u32x4 sq[4], so[4];
    auto stage_load = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;  // 1024 sixteen-byte chunks each
            const int rq = idx >> 5, cq = idx & 31;
            sq[i] = *reinterpret_cast<const u32x4*>(p.wq + (((int64_t)h * 32 + rq) * XC + cq * 8) * 2);
            const int ro = idx >> 2, po = idx & 3;
            so[i] = *reinterpret_cast<const u32x4*>(p.wo + ((int64_t)ro * XC + h * 32 + po * 8) * 2);
        }
    };
    auto stage_store = [&](uint8_t* st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 256 * i;
            const int rq = idx >> 5, cq = idx & 31;
            *reinterpret_cast<u32x4*>(st + rq * WQ_ROWB + cq * 16) = sq[i];
            const int ro = idx >> 2, po = idx & 3;
            uint8_t* dst = st + WQ_BYTES + ro * WO_ROWB + po * 16;
            const u32x2 lo = {so[i][0], so[i][1]}, hi = {so[i][2], so[i][3]};
            *reinterpret_cast<u32x2*>(dst) = lo;
            *reinterpret_cast<u32x2*>(dst + 8) = hi;
        }
    };
    stage_load(0);
    for (int i = tid; i < XC; i += 256) lbo[i] = p.bo ? ld_elem<DT>(p.bo, i) : 0.f;

    // ---- x panel -> registers, LayerNorm in registers ----
    typename E::v8 xf[XKC];
    load_panel<DT, XKC>(xf, p.x, XC, rowend, row0, l31, half);
    if (p.gamma != nullptr) layernorm_panel<DT, XKC>(xf, p.gamma, p.beta, p.eps, l31, half);

    f32x16 yacc[XC / 32];
#pragma unroll
    for (int ct = 0; ct < XC / 32; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[ct][r] = 0.f;

    stage_store(smem);
    __syncthreads();

    const uint8_t* k1b = p.k1 + (int64_t)bb * p.L1 * XC * 2;
    const uint8_t* v1b = p.v1t + (int64_t)bb * XH * XD * p.Lpad1 * 2;
    const float* bias1 = p.bias1 ? p.bias1 + (int64_t)bb * p.L1 : nullptr;
    const uint8_t* k2b = DUAL ? p.k2 + (int64_t)bb * p.L2 * XC * 2 : nullptr;
    const uint8_t* v2b = DUAL ? p.v2t + (int64_t)bb * XH * XD * p.Lpad2 * 2 : nullptr;

    for (int h = 0; h < XH; ++h) {
        const uint8_t* st = smem + (h & 1) * XSTAGE;
        stage_load(h + 1 < XH ? h + 1 : XH - 1);  // the last head re-loads itself: no divergent path
        XaFrags<DT> f1, f2;
        xa_prefetch<DT>(f1, k1b + h * XD * 2, v1b + (int64_t)h * XD * p.Lpad1 * 2, p.L1, p.Lpad1, l31, half);
        if (DUAL) xa_prefetch<DT>(f2, k2b + h * XD * 2, v2b + (int64_t)h * XD * p.Lpad2 * 2, p.L2, p.Lpad2, l31, half);

        // ---- q_h^T [32 dims x 32 tokens] = Wq_h . x^T ----
        f32x16 qt;
#pragma unroll
        for (int r = 0; r < 16; ++r) qt[r] = 0.f;
        {
            const uint8_t* wt = st + l31 * WQ_ROWB + half * 16;
            typename E::v8 wfa[4][1], wfb[4][1];
            rp_load_group<DT, XKC>(wfa, wt, 0);
#pragma unroll
            for (int g = 0; g < XKC / 4; g += 2) {
                rp_load_group<DT, XKC>(wfb, wt, (g + 1) * 4);
                rp_pin<DT, XKC>(wfa);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) qt = E::mfma32(wfa[cc][0], xf[g * 4 + cc], qt);
                if (g + 2 < XKC / 4) rp_load_group<DT, XKC>(wfa, wt, (g + 2) * 4);
                rp_pin<DT, XKC>(wfb);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) qt = E::mfma32(wfb[cc][0], xf[(g + 1) * 4 + cc], qt);
            }
        }
        // the reference materialises q in the storage type; registers 0..7 / 8..15 are the two K = 16 steps over the head dim
        typename E::v8 qb[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) qb[r >> 3][r & 7] = (typename E::elem)qt[r];

        // ---- attention of head h: segment 1 (text / T5), optional segment 2 (audio), blend ----
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.f;
        float inv = 1.f;
        xa_segment<DT>(f1, p.L1, bias1, p.scale_log2, qb, o, inv, half);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= inv;
        if (DUAL) {
            f32x16 o2;
#pragma unroll
            for (int r = 0; r < 16; ++r) o2[r] = 0.f;
            float inv2 = 1.f;
            xa_segment<DT>(f2, p.L2, nullptr, p.scale_log2, qb, o2, inv2, half);
            // text + ap_scale * audio (:454) in fp32, rounded once when packed for the output projection (the un-fused
            // path rounds each branch to the storage type first; emulating that cost ~100 VALU instructions per head in a
            // kernel whose instruction stream is 80 % vector ALU: 84 -> 75 us)
            const float s2 = p.scale2 * inv2;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = __builtin_fmaf(s2, o2[r], o[r]);
        }
        typename E::v8 ob[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[r >> 3][r & 7] = (typename E::elem)o[r];

        // ---- out^T [C x 32 tokens] += Wo[:, h*32 .. +32] . O_h^T  (A = Wo rows, k = head dim in the C-layout order) ----
        {
            const uint8_t* wo_t = st + WQ_BYTES + l31 * WO_ROWB + half * 8;
            u32x2 wlo[XC / 32][2], whi[XC / 32][2];
#pragma unroll
            for (int ct = 0; ct < XC / 32; ++ct)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    wlo[ct][kk] = *reinterpret_cast<const u32x2*>(wo_t + ct * 32 * WO_ROWB + kk * 32);
                    whi[ct][kk] = *reinterpret_cast<const u32x2*>(wo_t + ct * 32 * WO_ROWB + kk * 32 + 16);
                }
#pragma unroll
            for (int ct = 0; ct < XC / 32; ++ct)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    typename E::v8 wf = as_v8<DT>(make_uint4(wlo[ct][kk][0], wlo[ct][kk][1], whi[ct][kk][0], whi[ct][kk][1]));
                    yacc[ct] = E::mfma32(wf, ob[kk], yacc[ct]);
                }
        }
        stage_store(smem + ((h + 1) & 1) * XSTAGE);
        __syncthreads();
    }

    // ---- epilogue: out = y + b_out + x (raw) through the per-wave transpose scratch ----
    if (active) {
#pragma unroll
        for (int ct = 0; ct < XC / 32; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(lbo + ct * 32 + 8 * g + 4 * half);
                typename E::v4 y;
                y[0] = (typename E::elem)(yacc[ct][4 * g + 0] + b4.x);
                y[1] = (typename E::elem)(yacc[ct][4 * g + 1] + b4.y);
                y[2] = (typename E::elem)(yacc[ct][4 * g + 2] + b4.z);
                y[3] = (typename E::elem)(yacc[ct][4 * g + 3] + b4.w);
                *reinterpret_cast<uint2*>(scr + l31 * SCR_ROWB + (8 * g + 4 * half) * 2) = __builtin_bit_cast(uint2, y);
            }
            scratch_flush<DT>(scr, 32, p.out, XC, ct * 32, p.x, XC, row0, rowend, lane);
        }
    }
}

template <int DT, bool DUAL> int xa_launch(const XaP& p, hipStream_t s) {
    const size_t lds = 2 * XSTAGE + 4 * SCR_BYTES + XC * sizeof(float);
    auto kern = xattn_kernel<DT, DUAL>;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int64_t panels = (int64_t)p.B * ((p.N + 31) / 32);
    hipLaunchKernelGGL(kern, dim3((unsigned)((panels + 3) / 4)), dim3(256), lds, s, p);
    return apad_check_launch("apad_fused_cross_attention");
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int apad_fused_cross_attention(const apad_xattn_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_fused_cross_attention: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_fused_cross_attention: dtype %d not supported", d->dtype);
    if (d->C != XC || d->heads != XH || d->L1 > 64 || d->L2 > 64) {
        apad_set_error("apad_fused_cross_attention: C=%d heads=%d L1=%d L2=%d outside the kernel envelope (C 256, 8 heads, <= 64 keys)",
                       d->C, d->heads, d->L1, d->L2);
        return -3;
    }
    APAD_CHECK(d->x && d->wq && d->wo && d->k1 && d->v1t && d->out, "apad_fused_cross_attention: null operand");
    APAD_CHECK(d->B > 0 && d->N > 0 && d->L1 > 0 && d->Lpad1 >= d->L1 && d->Lpad1 % 32 == 0,
               "apad_fused_cross_attention: bad geometry B=%d N=%d L1=%d Lpad1=%d", d->B, d->N, d->L1, d->Lpad1);
    const bool dual = d->L2 > 0;
    if (dual)
        APAD_CHECK(d->k2 && d->v2t && d->Lpad2 >= d->L2 && d->Lpad2 % 32 == 0, "apad_fused_cross_attention: segment 2 needs k2 / v2t");
    APAD_CHECK((d->ln_gamma == nullptr) == (d->ln_beta == nullptr), "apad_fused_cross_attention: LayerNorm needs gamma and beta");
    APAD_CHECK(al16(d->x) && al16(d->wq) && al16(d->wo) && al16(d->k1) && al16(d->v1t) && al16(d->out) && al16(d->k2) && al16(d->v2t) &&
                   al16(d->ln_gamma) && al16(d->ln_beta),
               "apad_fused_cross_attention: pointers must be 16-byte aligned");
    XaP p;
    p.x = (const uint8_t*)d->x; p.gamma = (const uint8_t*)d->ln_gamma; p.beta = (const uint8_t*)d->ln_beta;
    p.wq = (const uint8_t*)d->wq; p.wo = (const uint8_t*)d->wo; p.bo = (const uint8_t*)d->bo;
    p.k1 = (const uint8_t*)d->k1; p.v1t = (const uint8_t*)d->v1t; p.bias1 = d->key_bias;
    p.k2 = (const uint8_t*)d->k2; p.v2t = (const uint8_t*)d->v2t; p.out = (uint8_t*)d->out;
    p.B = d->B; p.N = d->N; p.L1 = d->L1; p.Lpad1 = d->Lpad1; p.L2 = d->L2; p.Lpad2 = d->Lpad2;
    p.eps = d->ln_eps; p.scale_log2 = d->softmax_scale * XA_LOG2E; p.scale2 = d->scale2;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == APAD_BF16) return dual ? xa_launch<APAD_BF16, true>(p, s) : xa_launch<APAD_BF16, false>(p, s);
    return dual ? xa_launch<APAD_F16, true>(p, s) : xa_launch<APAD_F16, false>(p, s);
}
