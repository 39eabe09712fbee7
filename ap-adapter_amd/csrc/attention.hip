// apad_attention: softmax(Q K^T * scale + bias) V, optionally decoupled into two independently normalised key
// segments blended as seg1 + scale2 * seg2 -- the arithmetic of IPAttnProcessor2_0 (text branch over the frozen
// to_k/to_v, audio branch over to_k_ip/to_v_ip, reference attention_processor.py:429-454) in ONE kernel that
// shares Q, keeps both softmaxes in registers and writes the blended heads once.
//
// Mapping (wave64, MFMA 32x32x16, fp32 accumulate):
//   * one wave owns 32 queries of one (batch, head); 4 waves per workgroup -> 128 queries per workgroup
//   * scores are computed TRANSPOSED, S^T = K . Q^T, so that after the MFMA each lane holds 16 keys of ONE
//     query (column = lane&31): the softmax max/sum are in-lane reductions plus one cross-half exchange
//   * P^T stays in registers: the C-layout of S^T (keys (r&3)+8*(r>>2)+4*half) is re-used directly as the
//     B operand of O^T = V^T . P^T by loading V^T with the SAME key permutation (two 8-byte pieces per lane),
//     so no LDS round trip and no cross-lane shuffle sits between the two MFMAs
//   * K rows (A operand of S^T) and V^T rows (A operand of O^T) are 16 B / 8 B contiguous loads; K/V of one
//     (batch, head) are <= 66 KB (<= 520 audio+text keys) and stay L2/L1 resident across the query tiles
//   * online softmax over 32-key tiles in the exp2 domain (scale*log2e folded into the scores)
// V must be supplied transposed per head, [Bk][H][D][Lpad] with zero padding (apad_gemm APAD_OUT_VT).
#include "common.h"

namespace {

struct AttnP {
    const uint8_t* q;
    const uint8_t* k;
    const uint8_t* vt;
    const uint8_t* k2;
    const uint8_t* vt2;
    uint8_t* out;
    const float* key_bias;
    int64_t q_sb, q_sn, k_sb, k_sl, vt_sb, k2_sb, k2_sl, vt2_sb, o_sb, o_sn;
    int32_t B, N, H, L, Lpad, L2, Lpad2, kvdiv, kvdiv2;
    float scale_log2, scale2;
};

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

// One softmax segment: accumulates un-normalised O^T into o[] and the per-lane partial row sum into lsum;
// m is the running max (log2 domain).  kbase/vbase point at this (batch, head)'s K rows / V^T rows.
template <int DT, int D>
__device__ __forceinline__ void segment(const uint8_t* kbase, int64_t k_sl, const uint8_t* vbase, int L, int Lpad,
                                        const float* bias, float scale_log2, const typename ET<DT>::v8* qf,
                                        f32x16* o, float& m, float& lsum, int l31, int half) {
    using E = ET<DT>;
    constexpr int KC = D / 16;         // 16-wide chunks of the head dim (QK^T reduction)
    constexpr int DT_TILES = (D + 31) / 32;  // 32-row tiles of V^T (output head-dim rows)
    const int ntiles = (L + 31) >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int key0 = t * 32;
        // ---- S^T tile = K[key0..key0+32) . Q^T ----
        int krow = key0 + l31;
        krow = krow < L ? krow : L - 1;
        const uint8_t* kp = kbase + ((int64_t)krow * k_sl + half * 8) * 2;
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            typename E::v8 kf = as_v8<DT>(*reinterpret_cast<const uint4*>(kp + c * 32));
            s = E::mfma32(kf, qf[c], s);
        }
        // ---- scale, bias, mask, tile max ----
        float tmax = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v = s[r] * scale_log2;
            if (bias) v += bias[key < L ? key : L - 1] * LOG2E;
            v = key < L ? v : NEG_BIG;
            s[r] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mnew);
        m = mnew;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f(s[r] - mnew);
            s[r] = pv;
            psum += pv;
        }
        lsum = lsum * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // ---- O^T += V^T . P^T : two K=16 steps; B operand = P^T in its C-layout key order ----
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            typename E::v8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (typename E::elem)s[st * 8 + j];
            const int kcol = key0 + st * 16 + 4 * half;  // keys kcol..kcol+3 and kcol+8..kcol+11
#pragma unroll
            for (int dt = 0; dt < DT_TILES; ++dt) {
                int drow = dt * 32 + l31;
                drow = drow < D ? drow : D - 1;
                const uint8_t* vp = vbase + ((int64_t)drow * Lpad + kcol) * 2;
                uint2 v0 = *reinterpret_cast<const uint2*>(vp);
                uint2 v1 = *reinterpret_cast<const uint2*>(vp + 16);
                typename E::v8 vf = as_v8<DT>(make_uint4(v0.x, v0.y, v1.x, v1.y));
                o[dt] = E::mfma32(vf, pf, o[dt]);
            }
        }
    }
}

template <int DT, int D, bool DUAL>
__global__ __launch_bounds__(256) void attn_kernel(AttnP p) {
    using E = ET<DT>;
    constexpr int KC = D / 16;
    constexpr int DT_TILES = (D + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    if (q0 >= p.N) return;
    int qi = q0 + l31;
    const bool qvalid = qi < p.N;
    qi = qvalid ? qi : p.N - 1;

    // Q^T B-operand fragments: lane holds Q[qi][c*16 + half*8 .. +8)
    typename E::v8 qf[KC];
    const uint8_t* qp = p.q + ((int64_t)b * p.q_sb + (int64_t)qi * p.q_sn + h * D + half * 8) * 2;
#pragma unroll
    for (int c = 0; c < KC; ++c) qf[c] = as_v8<DT>(*reinterpret_cast<const uint4*>(qp + c * 32));

    f32x16 o[DT_TILES];
#pragma unroll
    for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = NEG_BIG, lsum = 0.f;

    {
        const int bk = b / p.kvdiv;
        const uint8_t* kbase = p.k + ((int64_t)bk * p.k_sb + h * D) * 2;
        const uint8_t* vbase = p.vt + ((int64_t)bk * p.vt_sb + (int64_t)h * D * p.Lpad) * 2;
        const float* bias = p.key_bias ? p.key_bias + (int64_t)b * p.L : nullptr;
        segment<DT, D>(kbase, p.k_sl, vbase, p.L, p.Lpad, bias, p.scale_log2, qf, o, m, lsum, l31, half);
    }
    lsum += __shfl_xor(lsum, 32, 64);
    float inv = 1.0f / lsum;
#pragma unroll
    for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= inv;

    if (DUAL) {
        if (p.L2 > 0) {
            f32x16 o2[DT_TILES];
#pragma unroll
            for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o2[dt][r] = 0.f;
            float m2 = NEG_BIG, l2 = 0.f;
            const int bk = b / p.kvdiv2;
            const uint8_t* kbase = p.k2 + ((int64_t)bk * p.k2_sb + h * D) * 2;
            const uint8_t* vbase = p.vt2 + ((int64_t)bk * p.vt2_sb + (int64_t)h * D * p.Lpad2) * 2;
            segment<DT, D>(kbase, p.k2_sl, vbase, p.L2, p.Lpad2, nullptr, p.scale_log2, qf, o2, m2, l2, l31, half);
            l2 += __shfl_xor(l2, 32, 64);
            const float inv2 = 1.0f / l2;
            // the un-fused reference rounds each branch, and scale * audio, to the storage type before the add
#pragma unroll
            for (int dt = 0; dt < DT_TILES; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = (float)(typename E::elem)o[dt][r];
                    const float a = (float)(typename E::elem)(o2[dt][r] * inv2);
                    o[dt][r] = t + (float)(typename E::elem)(p.scale2 * a);
                }
        }
    }

    // ---- store: lane owns query qi; regs 4g..4g+3 are 4 consecutive head-dim columns ----
    if (!qvalid) return;
    uint8_t* op = p.out + ((int64_t)b * p.o_sb + (int64_t)qi * p.o_sn + h * D) * 2;
#pragma unroll
    for (int dt = 0; dt < DT_TILES; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dcol = dt * 32 + 8 * g + 4 * half;
            if (dcol < D) {
                typename E::v4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (typename E::elem)o[dt][g * 4 + j];
                *reinterpret_cast<uint2*>(op + dcol * 2) = __builtin_bit_cast(uint2, pk);
            }
        }
    }
}

template <int DT, int D> int launch_d(const AttnP& p, bool dual, dim3 grid, hipStream_t s) {
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

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int apad_attention(const apad_attn_desc* d, void* stream) {
    APAD_CHECK(d != nullptr, "apad_attention: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_attention: dtype %d not supported", d->dtype);
    APAD_CHECK(d->q && d->k && d->vt && d->out, "apad_attention: null operand");
    APAD_CHECK(d->B > 0 && d->N > 0 && d->H > 0 && d->L > 0, "apad_attention: empty problem B=%d N=%d H=%d L=%d", d->B, d->N,
               d->H, d->L);
    APAD_CHECK(d->Lpad >= d->L && d->Lpad % 32 == 0, "apad_attention: Lpad must be >= L and a multiple of 32");
    APAD_CHECK(d->kv_batch_div >= 1, "apad_attention: kv_batch_div must be >= 1");
    APAD_CHECK(al16(d->q) && al16(d->k) && al16(d->vt) && al16(d->out) && al16(d->k2) && al16(d->vt2),
               "apad_attention: pointers must be 16-byte aligned");
    APAD_CHECK(d->q_stride_n % 8 == 0 && d->q_stride_b % 8 == 0 && d->k_stride_l % 8 == 0 && d->k_stride_b % 8 == 0 &&
                   d->o_stride_n % 4 == 0 && d->o_stride_b % 4 == 0 && d->vt_stride_b % 8 == 0,
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
    p.q_sb = d->q_stride_b; p.q_sn = d->q_stride_n; p.k_sb = d->k_stride_b; p.k_sl = d->k_stride_l; p.vt_sb = d->vt_stride_b;
    p.k2_sb = d->k2_stride_b; p.k2_sl = d->k2_stride_l; p.vt2_sb = d->vt2_stride_b; p.o_sb = d->o_stride_b; p.o_sn = d->o_stride_n;
    p.B = d->B; p.N = d->N; p.H = d->H; p.L = d->L; p.Lpad = d->Lpad; p.L2 = d->L2; p.Lpad2 = d->Lpad2;
    p.kvdiv = d->kv_batch_div; p.kvdiv2 = dual ? d->kv2_batch_div : 1;
    p.scale_log2 = d->softmax_scale * 1.4426950408889634f;
    p.scale2 = d->scale2;
    dim3 grid((unsigned)((d->N + 127) / 128), (unsigned)d->H, (unsigned)d->B);
    hipStream_t s = (hipStream_t)stream;
    return d->dtype == APAD_BF16 ? launch_dt<APAD_BF16>(p, d->D, dual, grid, s) : launch_dt<APAD_F16>(p, d->D, dual, grid, s);
}
