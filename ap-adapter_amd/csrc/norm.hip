// LayerNorm (token-major rows) and GroupNorm(+SiLU) over NHWC activations.  HBM-bound: 16 B/lane loads, fp32
// statistics, one pass over the data per kernel.
#include "common.h"

namespace {

// ---------------- LayerNorm: one wave per row, row held in registers ----------------
template <int DT, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const uint8_t* x, const uint8_t* gamma, const uint8_t* beta,
                                                        uint8_t* out, int64_t M, int C, int64_t ldx, int64_t ldo, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nvec = C >> 3;
    float v[NV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
            unpack8<DT>(*reinterpret_cast<const uint4*>(x + (row * ldx + vi * 8) * 2), v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += v[i][e];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = lane + 64 * i;
        if (vi < nvec) {
            float g[8], b[8], y[8];
            unpack8<DT>(*reinterpret_cast<const uint4*>(gamma + vi * 16), g);
            unpack8<DT>(*reinterpret_cast<const uint4*>(beta + vi * 16), b);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            *reinterpret_cast<uint4*>(out + (row * ldo + vi * 8) * 2) = pack8<DT>(y);
        }
    }
}

// ---------------- GroupNorm statistics: one workgroup per (group, batch) ----------------
// x [B][HW][C], channels of group g are the contiguous slice [g*cg, (g+1)*cg), cg % 4 == 0.
template <int DT>
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint8_t* x, float* ws, int HW, int C, int G) {
    __shared__ float red[2][4];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = C / G, q = cg >> 2;  // 4-element (8-byte) pieces per pixel
    const uint8_t* base = x + ((int64_t)b * HW * C + (int64_t)g * cg) * 2;
    const float shift = ld_elem<DT>(base, 0);
    float s = 0.f, ss = 0.f;
    const int total = HW * q;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int px = idx / q, pc = idx - px * q;
        uint2 u = *reinterpret_cast<const uint2*>(base + ((int64_t)px * C + pc * 4) * 2);
        typename ET<DT>::v4 v = __builtin_bit_cast(typename ET<DT>::v4, u);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = (float)v[e] - shift;
            s += d;
            ss += d * d;
        }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = s;
        red[1][wave] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = (float)HW * (float)cg;
        const float ts = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        const float tss = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        const float md = ts / n;  // mean of (x - shift)
        float var = tss / n - md * md;
        var = var > 0.f ? var : 0.f;
        ws[((int64_t)b * G + g) * 2 + 0] = md + shift;
        ws[((int64_t)b * G + g) * 2 + 1] = var;
    }
}

template <int DT, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint8_t* x, const float* ws, const uint8_t* gamma,
                                                       const uint8_t* beta, uint8_t* out, int64_t nvec_total, int HW, int C,
                                                       int G, float eps) {
    const int cg = C / G, vpr = C >> 3;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < nvec_total; idx += stride) {
        const int64_t pix = idx / vpr;
        const int vc = (int)(idx - pix * vpr);
        const int64_t b = pix / HW;
        const int c0 = vc * 8;
        float v[8], g[8], bt[8], y[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(x + idx * 16), v);
        unpack8<DT>(*reinterpret_cast<const uint4*>(gamma + c0 * 2), g);
        unpack8<DT>(*reinterpret_cast<const uint4*>(beta + c0 * 2), bt);
        const int g0 = c0 / cg, g1 = (c0 + 4) / cg;
        const float m0 = ws[(b * G + g0) * 2], r0 = rsqrtf(ws[(b * G + g0) * 2 + 1] + eps);
        const float m1 = ws[(b * G + g1) * 2], r1 = rsqrtf(ws[(b * G + g1) * 2 + 1] + eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float mean = e < 4 ? m0 : m1, rstd = e < 4 ? r0 : r1;
            float t = (v[e] - mean) * rstd * g[e] + bt[e];
            if (SILU) {
                // the un-fused reference rounds the normalised tensor to the storage type before SiLU
                t = (float)(typename ET<DT>::elem)t;
                t = silu_f(t);
            }
            y[e] = t;
        }
        *reinterpret_cast<uint4*>(out + idx * 16) = pack8<DT>(y);
    }
}

template <int DT> int ln_launch(const void* x, const void* gamma, const void* beta, void* out, int64_t M, int C, int64_t ldx,
                                int64_t ldo, float eps, hipStream_t s) {
    dim3 grid((unsigned)((M + 3) / 4));
    const int nvec = C / 8;
    if (nvec <= 64)
        hipLaunchKernelGGL((layernorm_kernel<DT, 1>), grid, dim3(256), 0, s, (const uint8_t*)x, (const uint8_t*)gamma,
                           (const uint8_t*)beta, (uint8_t*)out, M, C, ldx, ldo, eps);
    else if (nvec <= 128)
        hipLaunchKernelGGL((layernorm_kernel<DT, 2>), grid, dim3(256), 0, s, (const uint8_t*)x, (const uint8_t*)gamma,
                           (const uint8_t*)beta, (uint8_t*)out, M, C, ldx, ldo, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<DT, 4>), grid, dim3(256), 0, s, (const uint8_t*)x, (const uint8_t*)gamma,
                           (const uint8_t*)beta, (uint8_t*)out, M, C, ldx, ldo, eps);
    return apad_check_launch("apad_layernorm");
}

template <int DT> int gn_launch(const void* x, const void* gamma, const void* beta, void* out, float* ws, int B, int HW, int C,
                                int G, float eps, int silu, hipStream_t s) {
    hipLaunchKernelGGL((gn_stats_kernel<DT>), dim3(G, B), dim3(256), 0, s, (const uint8_t*)x, ws, HW, C, G);
    int rc = apad_check_launch("apad_groupnorm(stats)");
    if (rc) return rc;
    const int64_t nvec = (int64_t)B * HW * (C / 8);
    int64_t blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (silu)
        hipLaunchKernelGGL((gn_apply_kernel<DT, true>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)x, ws,
                           (const uint8_t*)gamma, (const uint8_t*)beta, (uint8_t*)out, nvec, HW, C, G, eps);
    else
        hipLaunchKernelGGL((gn_apply_kernel<DT, false>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)x, ws,
                           (const uint8_t*)gamma, (const uint8_t*)beta, (uint8_t*)out, nvec, HW, C, G, eps);
    return apad_check_launch("apad_groupnorm(apply)");
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int apad_layernorm(const void* x, const void* gamma, const void* beta, void* out, int64_t M, int32_t C, int64_t ldx,
                              int64_t ldo, float eps, int32_t dtype, void* stream) {
    APAD_CHECK(x && gamma && beta && out, "apad_layernorm: null operand");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_layernorm: dtype %d not supported", dtype);
    APAD_CHECK(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "apad_layernorm: need M>0, C%%8==0, C<=2048 (M=%lld C=%d)", (long long)M, C);
    APAD_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && al16(x) && al16(out) && al16(gamma) && al16(beta),
               "apad_layernorm: rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    return dtype == APAD_BF16 ? ln_launch<APAD_BF16>(x, gamma, beta, out, M, C, ldx, ldo, eps, s)
                              : ln_launch<APAD_F16>(x, gamma, beta, out, M, C, ldx, ldo, eps, s);
}

extern "C" int64_t apad_groupnorm_workspace_bytes(int32_t B, int32_t G) { return (int64_t)B * G * 2 * sizeof(float); }

extern "C" int apad_groupnorm(const void* x, const void* gamma, const void* beta, void* out, void* workspace, int32_t B,
                              int32_t HW, int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype, void* stream) {
    APAD_CHECK(x && gamma && beta && out && workspace, "apad_groupnorm: null operand");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_groupnorm: dtype %d not supported", dtype);
    APAD_CHECK(B > 0 && HW > 0 && G > 0 && C % G == 0 && (C / G) % 4 == 0 && C % 8 == 0,
               "apad_groupnorm: need C%%G==0, (C/G)%%4==0, C%%8==0 (B=%d HW=%d C=%d G=%d)", B, HW, C, G);
    APAD_CHECK(al16(x) && al16(out) && al16(gamma) && al16(beta), "apad_groupnorm: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    return dtype == APAD_BF16 ? gn_launch<APAD_BF16>(x, gamma, beta, out, (float*)workspace, B, HW, C, G, eps, silu, s)
                              : gn_launch<APAD_F16>(x, gamma, beta, out, (float*)workspace, B, HW, C, G, eps, silu, s);
}
