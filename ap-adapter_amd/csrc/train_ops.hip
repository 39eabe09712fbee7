// Input-gradient kernels of the frozen UNet layers and the optimizer step of the adapter (SURVEY a-11; reference
// train_apadapter_v2.py:941-979: MSE in fp32, backward, clip-norm 1.0, AdamW).  The GEMM-shaped gradients (linear /
// convolution dgrad, adapter wgrad) reuse apad_gemm on transposed / flipped weights; this file holds what is left:
// LayerNorm / GroupNorm(+SiLU) / GEGLU backward, the data-movement duals of the strided and upsampled convolutions,
// a padded 2-D transpose for the wgrad GEMM, the loss, and the fused clip + AdamW update over the flat parameter buffer.
// All reductions run in a fixed order (no float atomics): a training step is bit-reproducible.
#include "common.h"

namespace {

// element-typed vector accessors: `e` is an ELEMENT offset from `base` (16-bit storage types and, for the fp32 training mode --
// the reference's default, train.sh leaves --mixed_precision unset -- float)
template <int DT> struct ESZ { static constexpr int v = 2; };
template <> struct ESZ<APAD_F32> { static constexpr int v = 4; };
template <int DT> __device__ __forceinline__ void ld4(const uint8_t* base, int64_t e, float* f) {
    if constexpr (DT == APAD_F32) {
        const float4 v = *reinterpret_cast<const float4*>(base + e * 4);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
        typename ET<DT>::v4 v = __builtin_bit_cast(typename ET<DT>::v4, *reinterpret_cast<const uint2*>(base + e * 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = (float)v[j];
    }
}
template <int DT> __device__ __forceinline__ void st4(uint8_t* base, int64_t e, const float* f) {
    if constexpr (DT == APAD_F32) {
        *reinterpret_cast<float4*>(base + e * 4) = make_float4(f[0], f[1], f[2], f[3]);
    } else {
        typename ET<DT>::v4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (typename ET<DT>::elem)f[j];
        *reinterpret_cast<uint2*>(base + e * 2) = __builtin_bit_cast(uint2, v);
    }
}
template <int DT> __device__ __forceinline__ void ld8(const uint8_t* base, int64_t e, float* f) {
    if constexpr (DT == APAD_F32) {
        ld4<DT>(base, e, f);
        ld4<DT>(base, e + 4, f + 4);
    } else {
        unpack8<DT>(*reinterpret_cast<const uint4*>(base + e * 2), f);
    }
}
template <int DT> __device__ __forceinline__ void st8(uint8_t* base, int64_t e, const float* f) {
    if constexpr (DT == APAD_F32) {
        st4<DT>(base, e, f);
        st4<DT>(base, e + 4, f + 4);
    } else {
        *reinterpret_cast<uint4*>(base + e * 2) = pack8<DT>(f);
    }
}

// block-wide sum of two values in a fixed order (256 threads); result broadcast to every thread
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sh /* [2*4] */) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // sh may still be read from a previous call
    if (lane == 0) { sh[wave] = a; sh[4 + wave] = b; }
    __syncthreads();
    a = sh[0] + sh[1] + sh[2] + sh[3];
    b = sh[4] + sh[5] + sh[6] + sh[7];
}

// ---- LayerNorm backward (input gradient): one wave per row -----------------------------------------------------------
// dres (optional): the gradient that reaches x past the LayerNorm (the residual connection of a pre-norm sub-layer): added here,
// to the storage-rounded LayerNorm gradient, instead of by a separate accumulation kernel of the autograd engine
template <int DT> __global__ __launch_bounds__(256) void ln_bwd_kernel(const uint8_t* x, const uint8_t* gamma, const uint8_t* dy,
                                                                       const uint8_t* dres, uint8_t* dx, int64_t M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int64_t r0 = row * C;  // element offset of the row
    const int nch = C / 8;
    float s = 0.f, ss = 0.f;
    for (int c = lane; c < nch; c += 64) {
        float v[8];
        ld8<DT>(x, r0 + c * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += v[j]; ss += v[j] * v[j]; }
    }
    s = wave_sum(s); ss = wave_sum(ss);
    const float mean = s / C;
    const float rstd = rsqrtf(fmaxf(ss / C - mean * mean, 0.f) + eps);
    float sg = 0.f, sgx = 0.f;
    for (int c = lane; c < nch; c += 64) {
        float v[8], g[8], w[8];
        ld8<DT>(x, r0 + c * 8, v);
        ld8<DT>(dy, r0 + c * 8, g);
        ld8<DT>(gamma, c * 8, w);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float gg = g[j] * w[j]; sg += gg; sgx += gg * (v[j] - mean) * rstd; }
    }
    sg = wave_sum(sg) / C; sgx = wave_sum(sgx) / C;
    for (int c = lane; c < nch; c += 64) {
        float v[8], g[8], w[8], o[8];
        ld8<DT>(x, r0 + c * 8, v);
        ld8<DT>(dy, r0 + c * 8, g);
        ld8<DT>(gamma, c * 8, w);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (g[j] * w[j] - sg - (v[j] - mean) * rstd * sgx);
        if (dres != nullptr) {
            float r[8];
            ld8<DT>(dres, r0 + c * 8, r);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (float)(typename ET<DT>::elem)o[j] + r[j];
        }
        st8<DT>(dx, r0 + c * 8, o);
    }
}

// ---- GroupNorm (+SiLU) backward: one workgroup per (group, sample), three sweeps over its [HW][cg] slab --------------
template <int DT, bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_kernel(const uint8_t* x, const uint8_t* gamma, const uint8_t* beta, const uint8_t* dy,
                                                     uint8_t* dx, int HW, int C, int G, float eps) {
    __shared__ float sh[8];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cg = C / G, q4 = cg / 4;  // 4-channel pieces per pixel of this group
    const int64_t base = (int64_t)b * HW * C + g * cg;  // element offset of the (sample, group) slab
    const int items = HW * q4;
    const float n = (float)HW * cg;
    float s = 0.f, ss = 0.f;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int pix = i / q4, c4 = i - pix * q4;
        float v[4];
        ld4<DT>(x, base + (int64_t)pix * C + c4 * 4, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) { s += v[j]; ss += v[j] * v[j]; }
    }
    block_sum2(s, ss, sh);
    const float mean = s / n;
    const float rstd = rsqrtf(fmaxf(ss / n - mean * mean, 0.f) + eps);
    float sg = 0.f, sgx = 0.f;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int pix = i / q4, c4 = i - pix * q4;
        float v[4], d[4], w[4], bb[4];
        ld4<DT>(x, base + (int64_t)pix * C + c4 * 4, v);
        ld4<DT>(dy, base + (int64_t)pix * C + c4 * 4, d);
        ld4<DT>(gamma, g * cg + c4 * 4, w);
        ld4<DT>(beta, g * cg + c4 * 4, bb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (v[j] - mean) * rstd;
            float dz = d[j];
            if (SILU) {
                const float z = xh * w[j] + bb[j];
                const float sig = DT == APAD_F32 ? 1.0f / (1.0f + expf(-z)) : __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
                dz *= sig * (1.0f + z * (1.0f - sig));
            }
            const float gg = dz * w[j];
            sg += gg; sgx += gg * xh;
        }
    }
    block_sum2(sg, sgx, sh);
    sg /= n; sgx /= n;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int pix = i / q4, c4 = i - pix * q4;
        float v[4], d[4], w[4], bb[4], o[4];
        ld4<DT>(x, base + (int64_t)pix * C + c4 * 4, v);
        ld4<DT>(dy, base + (int64_t)pix * C + c4 * 4, d);
        ld4<DT>(gamma, g * cg + c4 * 4, w);
        ld4<DT>(beta, g * cg + c4 * 4, bb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xh = (v[j] - mean) * rstd;
            float dz = d[j];
            if (SILU) {
                const float z = xh * w[j] + bb[j];
                const float sig = DT == APAD_F32 ? 1.0f / (1.0f + expf(-z)) : __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
                dz *= sig * (1.0f + z * (1.0f - sig));
            }
            o[j] = rstd * (dz * w[j] - sg - xh * sgx);
        }
        st4<DT>(dx, base + (int64_t)pix * C + c4 * 4, o);
    }
}

// ---- GEGLU forward / backward on the stored projection [M][2N] (value | gate) ----------------------------------------
template <int DT> __global__ __launch_bounds__(256) void geglu_fwd_kernel(const uint8_t* proj, uint8_t* h, int64_t M, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // 8-element chunk index over [M][N]
    const int nch = N / 8;
    if (i >= M * nch) return;
    const int64_t m = i / nch;
    const int c = (int)(i - m * nch);
    float v[8], g[8], o[8];
    ld8<DT>(proj, m * 2 * N + c * 8, v);
    ld8<DT>(proj, m * 2 * N + N + c * 8, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j] * (DT == APAD_F32 ? 0.5f * g[j] * (1.0f + erff(g[j] * 0.70710678118654752440f)) : gelu_erf_f(g[j]));
    st8<DT>(h, m * N + c * 8, o);
}
template <int DT> __global__ __launch_bounds__(256) void geglu_bwd_kernel(const uint8_t* proj, const uint8_t* dh, uint8_t* dproj, int64_t M, int N) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nch = N / 8;
    if (i >= M * nch) return;
    const int64_t m = i / nch;
    const int c = (int)(i - m * nch);
    float v[8], g[8], d[8], dv[8], dg[8];
    ld8<DT>(proj, m * 2 * N + c * 8, v);
    ld8<DT>(proj, m * 2 * N + N + c * 8, g);
    ld8<DT>(dh, m * N + c * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float cdf = 0.5f * (1.0f + (DT == APAD_F32 ? erff(g[j] * 0.70710678118654752440f) : erf_as(g[j] * 0.70710678118654752440f)));
        const float pdf = 0.3989422804014327f * (DT == APAD_F32 ? expf(-0.5f * g[j] * g[j]) : __builtin_amdgcn_exp2f(-0.72134752044448170368f * g[j] * g[j]));  // exp(-g^2/2)/sqrt(2 pi)
        dv[j] = d[j] * g[j] * cdf;
        dg[j] = d[j] * v[j] * (cdf + g[j] * pdf);
    }
    st8<DT>(dproj, m * 2 * N + c * 8, dv);
    st8<DT>(dproj, m * 2 * N + N + c * 8, dg);
}

// ---- duals of the convolution gathers ---------------------------------------------------------------------------------
// nearest upsample backward: dx[b][h][w][c] = sum of dup[b][h'][w'][c] over the h', w' with floor(h'*H/Hup) == h (same map as
// the implicit-GEMM gather, gemm.hip)
template <int DT> __global__ __launch_bounds__(256) void upsample_bwd_kernel(const uint8_t* dup, uint8_t* dx, int B, int H, int W, int Hup, int Wup, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nch = C / 8;
    if (i >= (int64_t)B * H * W * nch) return;
    const int c = (int)(i % nch);
    int64_t r = i / nch;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    const int h0 = (h * Hup + H - 1) / H, h1 = ((h + 1) * Hup + H - 1) / H;  // h' in [ceil(h*Hup/H), ceil((h+1)*Hup/H))
    const int w0 = (w * Wup + W - 1) / W, w1 = ((w + 1) * Wup + W - 1) / W;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int hh = h0; hh < h1; ++hh)
        for (int ww = w0; ww < w1; ++ww) {
            float v[8];
            ld8<DT>(dup, (((int64_t)b * Hup + hh) * Wup + ww) * C + c * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
    st8<DT>(dx, (((int64_t)b * H + h) * W + w) * C + c * 8, acc);
}
// stride-2 convolution backward, step 1: z[b][2i][2j] = dy[b][i][j], zero elsewhere (z is [B][H][W][C]); the stride-1
// convolution of z with the flipped weights is then the input gradient
template <int DT> __global__ __launch_bounds__(256) void zero_stuff_kernel(const uint8_t* dy, uint8_t* z, int B, int H, int W, int Ho, int Wo, int C) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nch = C / 8;
    if (i >= (int64_t)B * H * W * nch) return;
    const int c = (int)(i % nch);
    int64_t r = i / nch;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int b = (int)(r / H);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((h & 1) == 0 && (w & 1) == 0 && (h >> 1) < Ho && (w >> 1) < Wo)
        ld8<DT>(dy, (((int64_t)b * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c * 8, v);
    st8<DT>(z, (((int64_t)b * H + h) * W + w) * C + c * 8, v);
}
// x [M][C] -> xt [C][Mpad] (columns m >= M zero): both operands of the adapter weight-gradient GEMM dW = dK^T . ehs
template <int DT> __global__ __launch_bounds__(256) void transpose_pad_kernel(const uint8_t* x, uint8_t* xt, int M, int C, int Mpad) {
    using E = ET<DT>;
    __shared__ typename E::elem tile[32][33];
    const int c0 = blockIdx.y * 32, m0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const typename E::elem* xin = reinterpret_cast<const typename E::elem*>(x);
    typename E::elem* xo = reinterpret_cast<typename E::elem*>(xt);
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, c = c0 + tx;
        tile[i][tx] = (m < M && c < C) ? xin[(int64_t)m * C + c] : (typename E::elem)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, m = m0 + tx;
        if (c < C && m < Mpad) xo[(int64_t)c * Mpad + m] = tile[tx][i];
    }
}

// two operands of ONE weight-gradient GEMM in one launch (blockIdx.z = operand), optionally widened to fp32 on the way (the
// exact-f32 MFMA path takes fp32 operands): replaces two transposes + two cast kernels per adapter tensor and micro-step
template <int DT, bool F32OUT>
__global__ __launch_bounds__(256) void transpose_pad2_kernel(const uint8_t* x0, uint8_t* xt0, int C0, const uint8_t* x1, uint8_t* xt1, int C1, int M,
                                                             int Mpad) {
    using E = ET<DT>;
    __shared__ typename E::elem tile[32][33];
    const bool second = blockIdx.z == 1;
    const int C = second ? C1 : C0;
    const int c0 = blockIdx.y * 32, m0 = blockIdx.x * 32;
    if (c0 >= C) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const typename E::elem* xin = reinterpret_cast<const typename E::elem*>(second ? x1 : x0);
    uint8_t* xo = second ? xt1 : xt0;
    for (int i = ty; i < 32; i += 8) {
        const int m = m0 + i, c = c0 + tx;
        tile[i][tx] = (m < M && c < C) ? xin[(int64_t)m * C + c] : (typename E::elem)0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, m = m0 + tx;
        if (c < C && m < Mpad) {
            if (F32OUT) reinterpret_cast<float*>(xo)[(int64_t)c * Mpad + m] = (float)tile[tx][i];
            else reinterpret_cast<typename E::elem*>(xo)[(int64_t)c * Mpad + m] = tile[tx][i];
        }
    }
}

// ---- loss ------------------------------------------------------------------------------------------------------------
// partial[blockIdx] = sum (pred - target)^2 over the block's slice; dpred = 2 (pred - target) / n  (F.mse_loss, mean)
// grad_scale: static loss scale, applied in fp32 BEFORE dpred is rounded to the storage type (2 (p - t) / n is ~1e-5 at full
// geometry: subnormal in f16)
template <int DT> __global__ __launch_bounds__(256) void mse_kernel(const uint8_t* pred, const float* target, uint8_t* dpred, float* partial, int64_t n,
                                                                    float grad_scale) {
    __shared__ float sh[8];
    float acc = 0.f, dummy = 0.f;
    const float k = 2.0f * grad_scale / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = ld_elem<DT>(pred, i) - target[i];
        acc += d * d;
        st_elem<DT>(dpred, i, d * k);
    }
    block_sum2(acc, dummy, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ __launch_bounds__(256) void final_sum_kernel(const float* partial, int nparts, float* out, float mul, int take_sqrt) {
    __shared__ float sh[8];
    float acc = 0.f, dummy = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partial[i];
    block_sum2(acc, dummy, sh);
    if (threadIdx.x == 0) out[0] = take_sqrt ? sqrtf(acc * mul) : acc * mul;
}

// ---- optimizer: global grad norm, then clip + AdamW on the flat fp32 master buffer (+ working copy in dtype) ----------
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, float* partial, int64_t n) {
    __shared__ float sh[8];
    float acc = 0.f, dummy = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc += g[i] * g[i];
    block_sum2(acc, dummy, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
template <int DT> __global__ __launch_bounds__(256) void adamw_kernel(float* param, uint8_t* work, const float* grad, float* m, float* v,
                                                                      const float* grad_norm, const int32_t* step, int64_t n, float lr,
                                                                      float beta1, float beta2, float eps, float wd, float max_norm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // a non-finite gradient norm (f16 overflow under loss scaling) skips the update, as accelerate's GradScaler does
    if (grad_norm != nullptr && !(fabsf(grad_norm[0]) < 3.0e38f)) return;
    // torch.nn.utils.clip_grad_norm_: coefficient = max_norm / (norm + 1e-6), clamped to 1
    float clip = 1.0f;
    if (max_norm > 0.f) clip = fminf(max_norm / (grad_norm[0] + 1e-6f), 1.0f);
    const int t = step[0];  // already advanced to the step being taken (>= 1)
    const float bc1 = 1.0f - __powf(beta1, (float)t), bc2 = 1.0f - __powf(beta2, (float)t);
    const float g = grad[i] * clip;
    float p = param[i] * (1.0f - lr * wd);  // decoupled weight decay (torch.optim.AdamW)
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;
    m[i] = mi; v[i] = vi;
    p -= (lr / bc1) * mi / (sqrtf(vi) / sqrtf(bc2) + eps);
    param[i] = p;
    if (work != nullptr) st_elem<DT>(work, i, p);
}

constexpr int REDUCE_BLOCKS = 1024;

}  // namespace

#define TRAIN_DT_CHECK(name) APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, name ": dtype %d not supported", dtype)
#define LAUNCH_DT(kern, grid, ...)                                                              \
    do {                                                                                        \
        if (dtype == APAD_BF16) hipLaunchKernelGGL((kern<APAD_BF16>), grid, dim3(256), 0, s, __VA_ARGS__); \
        else if (dtype == APAD_F32) hipLaunchKernelGGL((kern<APAD_F32>), grid, dim3(256), 0, s, __VA_ARGS__); \
        else hipLaunchKernelGGL((kern<APAD_F16>), grid, dim3(256), 0, s, __VA_ARGS__);          \
    } while (0)

extern "C" int apad_layernorm_bwd_add(const void* x, const void* gamma, const void* dy, const void* dres, void* dx, int64_t M, int32_t C,
                                      float eps, int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_layernorm_bwd");
    APAD_CHECK(x && gamma && dy && dx && M > 0 && C > 0 && C % 8 == 0, "apad_layernorm_bwd: bad operands (C %% 8 == 0 required)");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_DT(ln_bwd_kernel, dim3((unsigned)((M + 3) / 4)), (const uint8_t*)x, (const uint8_t*)gamma, (const uint8_t*)dy, (const uint8_t*)dres,
              (uint8_t*)dx, M, C, eps);
    return apad_check_launch("apad_layernorm_bwd");
}

extern "C" int apad_layernorm_bwd(const void* x, const void* gamma, const void* dy, void* dx, int64_t M, int32_t C, float eps,
                                  int32_t dtype, void* stream) {
    return apad_layernorm_bwd_add(x, gamma, dy, nullptr, dx, M, C, eps, dtype, stream);
}

extern "C" int apad_groupnorm_bwd(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, int32_t B, int32_t HW,
                                  int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_groupnorm_bwd");
    APAD_CHECK(x && gamma && beta && dy && dx && B > 0 && HW > 0 && G > 0 && C % G == 0 && (C / G) % 4 == 0,
               "apad_groupnorm_bwd: bad operands (channels per group must be a multiple of 4)");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((unsigned)G, (unsigned)B);
#define GN_ARGS (const uint8_t*)x, (const uint8_t*)gamma, (const uint8_t*)beta, (const uint8_t*)dy, (uint8_t*)dx, HW, C, G, eps
    if (dtype == APAD_BF16) {
        if (silu) hipLaunchKernelGGL((gn_bwd_kernel<APAD_BF16, true>), grid, dim3(256), 0, s, GN_ARGS);
        else hipLaunchKernelGGL((gn_bwd_kernel<APAD_BF16, false>), grid, dim3(256), 0, s, GN_ARGS);
    } else if (dtype == APAD_F32) {
        if (silu) hipLaunchKernelGGL((gn_bwd_kernel<APAD_F32, true>), grid, dim3(256), 0, s, GN_ARGS);
        else hipLaunchKernelGGL((gn_bwd_kernel<APAD_F32, false>), grid, dim3(256), 0, s, GN_ARGS);
    } else {
        if (silu) hipLaunchKernelGGL((gn_bwd_kernel<APAD_F16, true>), grid, dim3(256), 0, s, GN_ARGS);
        else hipLaunchKernelGGL((gn_bwd_kernel<APAD_F16, false>), grid, dim3(256), 0, s, GN_ARGS);
    }
#undef GN_ARGS
    return apad_check_launch("apad_groupnorm_bwd");
}

extern "C" int apad_geglu(const void* proj, void* h, int64_t M, int32_t N, int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_geglu");
    APAD_CHECK(proj && h && M > 0 && N > 0 && N % 8 == 0, "apad_geglu: bad operands");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_DT(geglu_fwd_kernel, dim3((unsigned)((M * (N / 8) + 255) / 256)), (const uint8_t*)proj, (uint8_t*)h, M, N);
    return apad_check_launch("apad_geglu");
}
extern "C" int apad_geglu_bwd(const void* proj, const void* dh, void* dproj, int64_t M, int32_t N, int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_geglu_bwd");
    APAD_CHECK(proj && dh && dproj && M > 0 && N > 0 && N % 8 == 0, "apad_geglu_bwd: bad operands");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_DT(geglu_bwd_kernel, dim3((unsigned)((M * (N / 8) + 255) / 256)), (const uint8_t*)proj, (const uint8_t*)dh, (uint8_t*)dproj, M, N);
    return apad_check_launch("apad_geglu_bwd");
}

extern "C" int apad_upsample_nearest_bwd(const void* dup, void* dx, int32_t B, int32_t H, int32_t W, int32_t Hup, int32_t Wup, int32_t C,
                                         int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_upsample_nearest_bwd");
    APAD_CHECK(dup && dx && B > 0 && H > 0 && W > 0 && Hup >= H && Wup >= W && C % 8 == 0, "apad_upsample_nearest_bwd: bad operands");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W * (C / 8);
    LAUNCH_DT(upsample_bwd_kernel, dim3((unsigned)((total + 255) / 256)), (const uint8_t*)dup, (uint8_t*)dx, B, H, W, Hup, Wup, C);
    return apad_check_launch("apad_upsample_nearest_bwd");
}
extern "C" int apad_zero_stuff2(const void* dy, void* z, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype,
                                void* stream) {
    TRAIN_DT_CHECK("apad_zero_stuff2");
    APAD_CHECK(dy && z && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && C % 8 == 0, "apad_zero_stuff2: bad operands");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * H * W * (C / 8);
    LAUNCH_DT(zero_stuff_kernel, dim3((unsigned)((total + 255) / 256)), (const uint8_t*)dy, (uint8_t*)z, B, H, W, Ho, Wo, C);
    return apad_check_launch("apad_zero_stuff2");
}
extern "C" int apad_transpose_pad(const void* x, void* xt, int32_t M, int32_t C, int32_t Mpad, int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_transpose_pad");
    APAD_CHECK(x && xt && M > 0 && C > 0 && Mpad >= M && Mpad % 32 == 0, "apad_transpose_pad: bad operands");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_DT(transpose_pad_kernel, dim3((unsigned)(Mpad / 32), (unsigned)((C + 31) / 32)), (const uint8_t*)x, (uint8_t*)xt, M, C, Mpad);
    return apad_check_launch("apad_transpose_pad");
}

extern "C" int apad_transpose_pad2(const void* x0, void* xt0, int32_t C0, const void* x1, void* xt1, int32_t C1, int32_t M, int32_t Mpad,
                                   int32_t dtype, int32_t out_f32, void* stream) {
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_transpose_pad2: dtype %d not supported (16-bit inputs)", dtype);
    APAD_CHECK(x0 && xt0 && x1 && xt1 && M > 0 && C0 > 0 && C1 > 0 && Mpad >= M && Mpad % 32 == 0, "apad_transpose_pad2: bad operands");
    hipStream_t s = (hipStream_t)stream;
    const int Cm = C0 > C1 ? C0 : C1;
    dim3 grid((unsigned)(Mpad / 32), (unsigned)((Cm + 31) / 32), 2);
#define TP2(DT_, F_) hipLaunchKernelGGL((transpose_pad2_kernel<DT_, F_>), grid, dim3(256), 0, s, (const uint8_t*)x0, (uint8_t*)xt0, C0, \
                                        (const uint8_t*)x1, (uint8_t*)xt1, C1, M, Mpad)
    if (dtype == APAD_BF16) { if (out_f32) TP2(APAD_BF16, true); else TP2(APAD_BF16, false); }
    else { if (out_f32) TP2(APAD_F16, true); else TP2(APAD_F16, false); }
#undef TP2
    return apad_check_launch("apad_transpose_pad2");
}

extern "C" int64_t apad_reduce_workspace_bytes(void) { return (int64_t)REDUCE_BLOCKS * sizeof(float); }

extern "C" int apad_mse_loss_grad(const void* pred, const float* target, void* dpred, float* loss, float* workspace, int64_t n,
                                  float grad_scale, int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_mse_loss_grad");
    APAD_CHECK(pred && target && dpred && loss && workspace && n > 0, "apad_mse_loss_grad: bad operands");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((n + 255) / 256 < REDUCE_BLOCKS ? (n + 255) / 256 : REDUCE_BLOCKS);
    LAUNCH_DT(mse_kernel, dim3((unsigned)blocks), (const uint8_t*)pred, target, (uint8_t*)dpred, workspace, n, grad_scale);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, blocks, loss, 1.0f / (float)n, 0);
    return apad_check_launch("apad_mse_loss_grad");
}

// step[0] += 1 unless the gradient norm is non-finite (the skipped step of a loss-scaled f16 run keeps its bias-correction index)
__global__ void step_advance_if_finite_kernel(int32_t* step, const float* norm) {
    if (norm == nullptr || fabsf(norm[0]) < 3.0e38f) step[0] += 1;
}
extern "C" int apad_step_advance_if_finite(int32_t* step, const float* grad_norm, void* stream) {
    APAD_CHECK(step != nullptr, "apad_step_advance_if_finite: null pointer");
    hipLaunchKernelGGL(step_advance_if_finite_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, grad_norm);
    return apad_check_launch("apad_step_advance_if_finite");
}

extern "C" int apad_grad_norm(const float* grad, float* norm, float* workspace, int64_t n, void* stream) {
    APAD_CHECK(grad && norm && workspace && n > 0, "apad_grad_norm: bad operands");
    hipStream_t s = (hipStream_t)stream;
    const int blocks = (int)((n + 255) / 256 < REDUCE_BLOCKS ? (n + 255) / 256 : REDUCE_BLOCKS);
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, grad, workspace, n);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, s, (const float*)workspace, blocks, norm, 1.0f, 1);
    return apad_check_launch("apad_grad_norm");
}

extern "C" int apad_adamw_step(float* param, void* work, const float* grad, float* exp_avg, float* exp_avg_sq, const float* grad_norm,
                               const int32_t* step, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                               float max_grad_norm, int32_t dtype, void* stream) {
    TRAIN_DT_CHECK("apad_adamw_step");
    APAD_CHECK(param && grad && exp_avg && exp_avg_sq && step && n > 0, "apad_adamw_step: bad operands");
    APAD_CHECK(max_grad_norm <= 0.f || grad_norm != nullptr, "apad_adamw_step: clipping needs the gradient norm");
    hipStream_t s = (hipStream_t)stream;
    LAUNCH_DT(adamw_kernel, dim3((unsigned)((n + 255) / 256)), param, (uint8_t*)work, grad, exp_avg, exp_avg_sq, grad_norm, step, n, lr,
              beta1, beta2, eps, weight_decay, max_grad_norm);
    return apad_check_launch("apad_adamw_step");
}
