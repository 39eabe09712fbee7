// Small HBM-bound kernels of the path: AudioMAE token pooling, sinusoidal timestep embedding, fused
// classifier-free-guidance + DDIM update, device-side step counter.
#include "common.h"
#include "f32_ops.h"

namespace {

// rep [B][513][768] -> out [B][(64/tp)*(8/fp)][768]; token (t,f) of the 64x8 grid is row 1 + 8*t + f.
// (avg + max) / 2 over (tp x fp) windows (reference AudioMAE.py:148-182).
template <int DT, int ODT>
__global__ __launch_bounds__(256) void pool_kernel(const uint8_t* rep, uint8_t* out, int B, int tp, int fp) {
    const int nt = 64 / tp, nf = 8 / fp, La = nt * nf;
    const int64_t total = (int64_t)B * La * 96;  // 96 vectors of 8 channels
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int vc = (int)(idx % 96);
        const int64_t tok = idx / 96;
        const int b = (int)(tok / La), o = (int)(tok % La);
        const int ot = o / nf, of = o % nf;
        float s[8], mx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[e] = 0.f;
            mx[e] = -3.0e38f;
        }
        for (int dt = 0; dt < tp; ++dt)
            for (int df = 0; df < fp; ++df) {
                const int row = 1 + 8 * (ot * tp + dt) + (of * fp + df);
                float v[8];
                unpack8<DT>(*reinterpret_cast<const uint4*>(rep + (((int64_t)b * 513 + row) * 768 + vc * 8) * 2), v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s[e] += v[e];
                    mx[e] = fmaxf(mx[e], v[e]);
                }
            }
        const float inv = 1.0f / (float)(tp * fp);
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (s[e] * inv + mx[e]) * 0.5f;
        if (ODT == APAD_F32) {
            float4* op = reinterpret_cast<float4*>(out + (tok * 768 + vc * 8) * 4);
            op[0] = make_float4(y[0], y[1], y[2], y[3]);
            op[1] = make_float4(y[4], y[5], y[6], y[7]);
        } else {
            constexpr int PD = (ODT == APAD_F32) ? APAD_BF16 : ODT;
            *reinterpret_cast<uint4*>(out + (tok * 768 + vc * 8) * 2) = pack8<PD>(y);
        }
    }
}

// diffusers get_timestep_embedding: exponent = -ln(10000) * i / (half - freq_shift); [sin | cos], flipped to
// [cos | sin] when flip_sin_to_cos.
template <int DT>
__global__ void timestep_kernel(const float* t, uint8_t* out, int n, int dim, int flip, float freq_shift) {
    const int half = dim >> 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * half) return;
    const int r = idx / half, i = idx - r * half;
    // fp32 mode: the exponent in f32 exactly as torch forms it, exp correctly rounded (via f64) -- at t ~ 1000 one ulp of the
    // frequency is 6e-5 in the sin / cos argument
    const float ex = (-9.210340371976184f * (float)i) / ((float)half - freq_shift);
    const float freq = DT == APAD_F32 ? (float)exp((double)ex) : expf(ex);
    const float a = t[r] * freq;
    const float sv = sinf(a), cv = cosf(a);
    const int64_t base = (int64_t)r * dim;
    if (flip) {
        st_elem<DT>(out, base + i, cv);
        st_elem<DT>(out, base + half + i, sv);
    } else {
        st_elem<DT>(out, base + i, sv);
        st_elem<DT>(out, base + half + i, cv);
    }
}

// eps = e_u + g (e_c - e_u); x_prev = c0 * x + c1 * eps   (pipeline_audioldm2.py:1020-1025, DDIM eta = 0)
template <int DT>
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const uint8_t* eps2, float* latents, uint8_t* unet_in, float* eps_out,
                                                       const float* coef, const int32_t* step_ptr, float gs, int64_t total) {
    const int step = step_ptr ? *step_ptr : 0;
    const float c0 = coef[2 * step], c1 = coef[2 * step + 1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float eu = ld_elem<DT>(eps2, i), ec = ld_elem<DT>(eps2, total + i);
        // the reference forms the guided noise in the model dtype
        const float e = (float)(typename ET<DT>::elem)(eu + gs * (ec - eu));
        const float x = c0 * latents[i] + c1 * e;
        latents[i] = x;
        st_elem<DT>(unet_in, i, x);
        if (eps_out) eps_out[i] = e;
    }
}

__global__ void step_advance_kernel(int32_t* p) { *p = *p + 1; }

template <int DT> __global__ __launch_bounds__(256) void mix3_kernel(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* out, int64_t n,
                                                                     float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        st_elem<DT>(out, i, (ld_elem<DT>(a, i) + ld_elem<DT>(b, i) + ld_elem<DT>(c, i)) * scale);
}

// out[m][n] = softmax_n(scale * x[m][n]); one wave per row, statistics and exponentials in fp32 (libm expf: the op runs once per
// decoded clip, not per denoise step).  Serves the VAE mid-block's single-head d = 512 attention, which lies outside
// apad_attention's head-dim envelope and runs as apad_gemm (Q.K^T) -> this -> apad_gemm (P.V).
template <int DT> __global__ __launch_bounds__(256) void softmax_rows_kernel(const uint8_t* x, const float* bias, uint8_t* out, int64_t M, int N,
                                                                             int64_t ldx, int64_t ldb, int64_t ldo, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int64_t xo = m * ldx, oo = m * ldo;
    const float* bm = bias ? bias + m * ldb : nullptr;
    auto at = [&](int n) { return ld_elem<DT>(x, xo + n) * scale + (bm ? bm[n] : 0.f); };
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, at(n));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (mx == -INFINITY) mx = 0.f;  // a fully masked row: exp(-inf) = 0 everywhere, written as zeros below
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) sum += expf(at(n) - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    for (int n = lane; n < N; n += 64) st_elem<DT>(out, oo + n, expf(at(n) - mx) * inv);
}

// T5LayerNorm (mode 0: x * rsqrt(mean(x^2) + eps) * gamma) and F.normalize (mode 1: x / max(||x||, eps)); one wave per row
template <int DT> __global__ __launch_bounds__(256) void rmsnorm_kernel(const uint8_t* x, const uint8_t* gamma, uint8_t* out, int64_t M, int C,
                                                                        int64_t ldx, int64_t ldo, float eps, int mode) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float v = ld_elem<DT>(x, m * ldx + c);
        ss = fmaf(v, v, ss);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float r = mode == 0 ? 1.0f / sqrtf(ss / (float)C + eps) : 1.0f / fmaxf(sqrtf(ss), eps);
    for (int c = lane; c < C; c += 64) st_elem<DT>(out, m * ldo + c, ld_elem<DT>(x, m * ldx + c) * r * (mode == 0 ? ld_elem<DT>(gamma, c) : 1.0f));
}

// nn.Embedding: one wave per output row
template <int DT> __global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* table, const int64_t* ids, uint8_t* out, int64_t n,
                                                                            int64_t rows, int C) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int64_t id = ids[i];
    const bool ok = id >= 0 && id < rows;
    for (int c = lane; c < C; c += 64) st_elem<DT>(out, i * C + c, ok ? ld_elem<DT>(table, id * C + c) : 0.f);
}

// DiagonalGaussianDistribution.sample() of the VAE encoder: moments [rows][2L] = (mean | logvar) per latent pixel ->
// out [rows][L] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale  (scale = the pipeline's scaling_factor, or 1)
template <int DT> __global__ __launch_bounds__(256) void gaussian_sample_kernel(const uint8_t* moments, const uint8_t* noise, uint8_t* out,
                                                                                int64_t rows, int L, float scale) {
    const int64_t n = rows * L;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / L;
        const int c = (int)(i - r * L);
        const float mean = ld_elem<DT>(moments, r * 2 * L + c);
        const float logvar = fminf(fmaxf(ld_elem<DT>(moments, r * 2 * L + L + c), -30.0f), 20.0f);
        st_elem<DT>(out, i, (mean + expf(0.5f * logvar) * ld_elem<DT>(noise, i)) * scale);
    }
}

}  // namespace

extern "C" int apad_audiomae_pool(const void* rep, void* out, int32_t B, int32_t tp, int32_t fp, int32_t dtype,
                                  int32_t out_dtype, void* stream) {
    APAD_CHECK(rep && out && B > 0, "apad_audiomae_pool: null operand / empty batch");
    if (dtype == APAD_F32) {
        APAD_CHECK(out_dtype == APAD_F32, "apad_audiomae_pool: f32 input needs f32 output");
        APAD_CHECK(tp > 0 && fp > 0 && 64 % tp == 0 && 8 % fp == 0, "apad_audiomae_pool: pooling (%d,%d) must divide (64,8)", tp, fp);
        return apad_f32_audiomae_pool(rep, out, B, tp, fp, (hipStream_t)stream);
    }
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_audiomae_pool: dtype %d not supported", dtype);
    APAD_CHECK(out_dtype == dtype || out_dtype == APAD_F32, "apad_audiomae_pool: out_dtype must equal dtype or be f32");
    APAD_CHECK(tp > 0 && fp > 0 && 64 % tp == 0 && 8 % fp == 0, "apad_audiomae_pool: pooling (%d,%d) must divide (64,8)", tp, fp);
    const int64_t total = (int64_t)B * (64 / tp) * (8 / fp) * 96;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* r = (const uint8_t*)rep;
    uint8_t* o = (uint8_t*)out;
    if (dtype == APAD_BF16) {
        if (out_dtype == APAD_F32)
            hipLaunchKernelGGL((pool_kernel<APAD_BF16, APAD_F32>), dim3((unsigned)blocks), dim3(256), 0, s, r, o, B, tp, fp);
        else
            hipLaunchKernelGGL((pool_kernel<APAD_BF16, APAD_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, r, o, B, tp, fp);
    } else {
        if (out_dtype == APAD_F32)
            hipLaunchKernelGGL((pool_kernel<APAD_F16, APAD_F32>), dim3((unsigned)blocks), dim3(256), 0, s, r, o, B, tp, fp);
        else
            hipLaunchKernelGGL((pool_kernel<APAD_F16, APAD_F16>), dim3((unsigned)blocks), dim3(256), 0, s, r, o, B, tp, fp);
    }
    return apad_check_launch("apad_audiomae_pool");
}

extern "C" int apad_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim, int32_t flip_sin_to_cos,
                                       float freq_shift, int32_t dtype, void* stream) {
    APAD_CHECK(t && out && n > 0 && dim > 0 && dim % 2 == 0, "apad_timestep_embedding: bad arguments");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_timestep_embedding: dtype %d not supported", dtype);
    const int total = n * (dim / 2);
    dim3 grid((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_F32)
        hipLaunchKernelGGL((timestep_kernel<APAD_F32>), grid, dim3(256), 0, s, t, (uint8_t*)out, n, dim, flip_sin_to_cos, freq_shift);
    else if (dtype == APAD_BF16)
        hipLaunchKernelGGL((timestep_kernel<APAD_BF16>), grid, dim3(256), 0, s, t, (uint8_t*)out, n, dim, flip_sin_to_cos, freq_shift);
    else
        hipLaunchKernelGGL((timestep_kernel<APAD_F16>), grid, dim3(256), 0, s, t, (uint8_t*)out, n, dim, flip_sin_to_cos, freq_shift);
    return apad_check_launch("apad_timestep_embedding");
}

extern "C" int apad_cfg_ddim_step(const void* eps2, float* latents, void* unet_in, float* eps_out, const float* coef,
                                  const int32_t* step_ptr, float guidance_scale, int32_t B, int64_t n, int32_t dtype,
                                  void* stream) {
    APAD_CHECK(eps2 && latents && unet_in && coef, "apad_cfg_ddim_step: null operand");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_cfg_ddim_step: dtype %d not supported", dtype);
    APAD_CHECK(B > 0 && n > 0, "apad_cfg_ddim_step: empty problem");
    const int64_t total = (int64_t)B * n;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_F32)
        hipLaunchKernelGGL((cfg_ddim_kernel<APAD_F32>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)eps2, latents,
                           (uint8_t*)unet_in, eps_out, coef, step_ptr, guidance_scale, total);
    else if (dtype == APAD_BF16)
        hipLaunchKernelGGL((cfg_ddim_kernel<APAD_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)eps2, latents,
                           (uint8_t*)unet_in, eps_out, coef, step_ptr, guidance_scale, total);
    else
        hipLaunchKernelGGL((cfg_ddim_kernel<APAD_F16>), dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)eps2, latents,
                           (uint8_t*)unet_in, eps_out, coef, step_ptr, guidance_scale, total);
    return apad_check_launch("apad_cfg_ddim_step");
}

extern "C" int apad_mix3(const void* a, const void* b, const void* c, void* out, int64_t n, float scale, int32_t dtype, void* stream) {
    APAD_CHECK(a && b && c && out && n > 0, "apad_mix3: bad operands");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_mix3: dtype %d not supported", dtype);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *pa = (const uint8_t*)a, *pb = (const uint8_t*)b, *pc = (const uint8_t*)c;
    if (dtype == APAD_F32) hipLaunchKernelGGL((mix3_kernel<APAD_F32>), dim3((unsigned)blocks), dim3(256), 0, s, pa, pb, pc, (uint8_t*)out, n, scale);
    else if (dtype == APAD_BF16) hipLaunchKernelGGL((mix3_kernel<APAD_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, pa, pb, pc, (uint8_t*)out, n, scale);
    else hipLaunchKernelGGL((mix3_kernel<APAD_F16>), dim3((unsigned)blocks), dim3(256), 0, s, pa, pb, pc, (uint8_t*)out, n, scale);
    return apad_check_launch("apad_mix3");
}

extern "C" int apad_softmax_rows(const void* x, const float* bias, void* out, int64_t M, int32_t N, int64_t ldx, int64_t ldb, int64_t ldo,
                                 float scale, int32_t dtype, void* stream) {
    APAD_CHECK(x && out && M > 0 && N > 0 && ldx >= N && ldo >= N && (!bias || ldb >= N),
               "apad_softmax_rows: bad operands (M=%lld N=%d ldx=%lld ldb=%lld ldo=%lld)", (long long)M, N, (long long)ldx, (long long)ldb, (long long)ldo);
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_softmax_rows: dtype %d not supported", dtype);
    const dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* px = (const uint8_t*)x;
    if (dtype == APAD_F32) hipLaunchKernelGGL((softmax_rows_kernel<APAD_F32>), grid, dim3(256), 0, s, px, bias, (uint8_t*)out, M, N, ldx, ldb, ldo, scale);
    else if (dtype == APAD_BF16) hipLaunchKernelGGL((softmax_rows_kernel<APAD_BF16>), grid, dim3(256), 0, s, px, bias, (uint8_t*)out, M, N, ldx, ldb, ldo, scale);
    else hipLaunchKernelGGL((softmax_rows_kernel<APAD_F16>), grid, dim3(256), 0, s, px, bias, (uint8_t*)out, M, N, ldx, ldb, ldo, scale);
    return apad_check_launch("apad_softmax_rows");
}

extern "C" int apad_rmsnorm(const void* x, const void* gamma, void* out, int64_t M, int32_t C, int64_t ldx, int64_t ldo, float eps, int32_t mode,
                            int32_t dtype, void* stream) {
    APAD_CHECK(x && out && M > 0 && C > 0 && ldx >= C && ldo >= C && (mode == 0 || mode == 1) && (mode == 1 || gamma), "apad_rmsnorm: bad operands");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_rmsnorm: dtype %d not supported", dtype);
    const dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *px = (const uint8_t*)x, *pg = (const uint8_t*)gamma;
    if (dtype == APAD_F32) hipLaunchKernelGGL((rmsnorm_kernel<APAD_F32>), grid, dim3(256), 0, s, px, pg, (uint8_t*)out, M, C, ldx, ldo, eps, mode);
    else if (dtype == APAD_BF16) hipLaunchKernelGGL((rmsnorm_kernel<APAD_BF16>), grid, dim3(256), 0, s, px, pg, (uint8_t*)out, M, C, ldx, ldo, eps, mode);
    else hipLaunchKernelGGL((rmsnorm_kernel<APAD_F16>), grid, dim3(256), 0, s, px, pg, (uint8_t*)out, M, C, ldx, ldo, eps, mode);
    return apad_check_launch("apad_rmsnorm");
}

extern "C" int apad_gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int64_t rows, int32_t C, int32_t dtype, void* stream) {
    APAD_CHECK(table && ids && out && n > 0 && rows > 0 && C > 0, "apad_gather_rows: bad operands");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_gather_rows: dtype %d not supported", dtype);
    const dim3 grid((unsigned)((n + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    const uint8_t* pt = (const uint8_t*)table;
    if (dtype == APAD_F32) hipLaunchKernelGGL((gather_rows_kernel<APAD_F32>), grid, dim3(256), 0, s, pt, ids, (uint8_t*)out, n, rows, C);
    else if (dtype == APAD_BF16) hipLaunchKernelGGL((gather_rows_kernel<APAD_BF16>), grid, dim3(256), 0, s, pt, ids, (uint8_t*)out, n, rows, C);
    else hipLaunchKernelGGL((gather_rows_kernel<APAD_F16>), grid, dim3(256), 0, s, pt, ids, (uint8_t*)out, n, rows, C);
    return apad_check_launch("apad_gather_rows");
}

extern "C" int apad_gaussian_sample(const void* moments, const void* noise, void* out, int64_t rows, int32_t latent, float scale,
                                    int32_t dtype, void* stream) {
    APAD_CHECK(moments && noise && out && rows > 0 && latent > 0, "apad_gaussian_sample: bad operands");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16 || dtype == APAD_F32, "apad_gaussian_sample: dtype %d not supported", dtype);
    int64_t blocks = (rows * latent + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
    const uint8_t *pm = (const uint8_t*)moments, *pn = (const uint8_t*)noise;
    if (dtype == APAD_F32) hipLaunchKernelGGL((gaussian_sample_kernel<APAD_F32>), dim3((unsigned)blocks), dim3(256), 0, s, pm, pn, (uint8_t*)out, rows, latent, scale);
    else if (dtype == APAD_BF16) hipLaunchKernelGGL((gaussian_sample_kernel<APAD_BF16>), dim3((unsigned)blocks), dim3(256), 0, s, pm, pn, (uint8_t*)out, rows, latent, scale);
    else hipLaunchKernelGGL((gaussian_sample_kernel<APAD_F16>), dim3((unsigned)blocks), dim3(256), 0, s, pm, pn, (uint8_t*)out, rows, latent, scale);
    return apad_check_launch("apad_gaussian_sample");
}

extern "C" int apad_step_advance(int32_t* step_ptr, void* stream) {
    APAD_CHECK(step_ptr, "apad_step_advance: null pointer");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_ptr);
    return apad_check_launch("apad_step_advance");
}

// ---- measurement probe (bench.py's `mfma_ceiling`): four independent v_mfma_f32_32x32x16_bf16 chains per wave, two waves per SIMD, no memory
//      traffic; operands zero (mode 0) or eight rotating register sets of pseudo-random bf16 in [-1, 1) (mode 1).  The dense rate this chip delivers
//      depends on the operand data (power management: tools/ubench/mfma_data.hip); the bench line states it next to the nominal peak. ----
namespace {
__device__ __forceinline__ uint32_t probe_hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* sink, int mode, int iters) {
    bf16x8_t a[8], b[8];
    for (int s = 0; s < 8; ++s)
        for (int i = 0; i < 8; ++i) {
            float va = 0.f, vb = 0.f;
            if (mode == 1) {
                va = (float)(probe_hash32(threadIdx.x * 131u + s * 17u + i) & 0xffff) / 32768.f - 1.f;
                vb = (float)(probe_hash32(threadIdx.x * 977u + s * 29u + i + 7u) & 0xffff) / 32768.f - 1.f;
            }
            a[s][i] = (__bf16)va;
            b[s][i] = (__bf16)vb;
        }
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[(s + c) & 7], acc[c], 0, 0, 0);
    }
    float t = 0.f;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) t += acc[c][r];
    if (t == 12345.678f) sink[0] = t;
}
}  // namespace

// launches the probe on 512 workgroups of 256 threads; returns the FLOPs of the launch through *flops (2 x 32 x 32 x 16 per MFMA)
extern "C" int apad_probe_mfma(void* sink, int32_t mode, int32_t iters, double* flops, void* stream) {
    APAD_CHECK(sink && iters > 0 && (mode == 0 || mode == 1), "apad_probe_mfma: bad arguments");
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, (float*)sink, mode, iters);
    if (flops) *flops = 512.0 * 4 * iters * 32 * 2.0 * 32 * 32 * 16;
    return apad_check_launch("apad_probe_mfma");
}
