// LayerNorm (token-major rows) and GroupNorm(+SiLU) over NHWC activations.  HBM-bound: 16 B/lane loads, fp32
// statistics, one pass over the data per kernel.
#include "common.h"
#include "f32_ops.h"

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

// ---------------- GroupNorm over NHWC x[B][HW][C] ----------------
// Pass 1 (gn_partial_kernel): workgroup = (pixel chunk, batch); thread (vc, py) owns ONE 16-byte channel vector vc and
// walks pixels py, py+PY, ... of the chunk, so every load instruction reads whole contiguous pixel rows; per-channel
// partial sums are folded to per-group sums through LDS and written as partial[b][chunk][g] = (sum, sumsq).
// Pass 2 (gn_apply_kernel): workgroup = (pixel chunk, batch); the first G threads reduce the chunk partials of their
// group to mean / rstd in LDS, then all threads normalise (+ optional SiLU) with 16-byte loads/stores.
constexpr int GN_CHUNK = 64;  // pixels per workgroup

// The input of a GroupNorm as up to two channel-concatenated sources (the up blocks' torch.cat([hidden, skip], 1) is never
// materialised): channels [0, Ca) from a [Ba][HW][Ca], [Ca, Ca + Cb) from b [Bb][HW][Cb]; sample index modulo the source's batch
// (a skip of the CFG-shared prefix is stored once for both halves of the batch).  Cb = 0: one source.
struct GnSrc {
    const uint8_t* a;
    const uint8_t* b;
    int Ca, Cb, Ba, Bb;
};
// address of the 16-byte vector that starts at channel c (c % 8 == 0, Ca % 8 == 0) of pixel px of sample bi
__device__ __forceinline__ const uint8_t* gn_vec(const GnSrc& s, int bi, int HW, int px, int c) {
    return c < s.Ca ? s.a + (((int64_t)(bi % s.Ba) * HW + px) * s.Ca + c) * 2
                    : s.b + (((int64_t)(bi % s.Bb) * HW + px) * s.Cb + (c - s.Ca)) * 2;
}

template <int DT>
__global__ __launch_bounds__(256) void gn_partial_kernel(GnSrc x, float* partial, int HW, int C, int G, int nchunk) {
    __shared__ float ls[2][2048];  // [sum|sumsq][thread * 8 + e]: per-thread channel partials (deterministic reduction)
    __shared__ float lc[2][1280];  // per-channel sums of this workgroup (C <= 1280)
    const int vpr = C >> 3;        // 16-byte vectors per pixel (<= 160)
    const int PY = 256 / vpr;      // pixel lanes
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int p0 = chunk * GN_CHUNK, p1 = min(p0 + GN_CHUNK, HW);
    const int tid = threadIdx.x;
    const int vc = tid % vpr, py = tid / vpr;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    if (py < PY) {
        const uint8_t* base = gn_vec(x, b, HW, 0, vc * 8);  // (a thread's vector lies in one source)
        const int64_t pstride = (vc * 8 < x.Ca ? x.Ca : x.Cb) * 2;
        for (int px = p0 + py; px < p1; px += PY) {
            float v[8];
            unpack8<DT>(*reinterpret_cast<const uint4*>(base + (int64_t)px * pstride), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s[e] += v[e];
                ss[e] += v[e] * v[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ls[0][tid * 8 + e] = s[e];
        ls[1][tid * 8 + e] = ss[e];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {  // fixed-order sum over the pixel lanes
        const int cv = c >> 3, e = c & 7;
        float a0 = 0.f, a1 = 0.f;
        for (int q = 0; q < PY; ++q) {
            a0 += ls[0][(q * vpr + cv) * 8 + e];
            a1 += ls[1][(q * vpr + cv) * 8 + e];
        }
        lc[0][c] = a0;
        lc[1][c] = a1;
    }
    __syncthreads();
    const int cg = C / G;
    if (tid < G) {
        float a0 = 0.f, a1 = 0.f;
        for (int c = tid * cg; c < (tid + 1) * cg; ++c) {
            a0 += lc[0][c];
            a1 += lc[1][c];
        }
        float* o = partial + (((int64_t)b * nchunk + chunk) * G + tid) * 2;
        o[0] = a0;
        o[1] = a1;
    }
}

// Between the passes (gn_finalize_kernel): wave (b, g) sums the chunk partials of its group ONCE, in a fixed order -> stats[b][g] = (mean, rstd).
// (Until round 6 every workgroup of the apply pass repeated that sum -- 63 dependent-latency loads in front of its first pixel at the
// 4000-pixel level, where the pass ran at 2.5 TB/s.)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* partial, float* stats, int BG, int G, int nchunk, float n, float eps) {
    // one WAVE per (sample, group): lane k takes chunks k, k + 64, ..., then a fixed butterfly (a thread per pair walking its 63 chunks was 63
    // dependent L2 latencies: 17.9 us in-step for 2048 pairs)
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= BG) return;
    const int b = i / G, g = i - b * G;
    const float* pp = partial + ((int64_t)b * nchunk * G + g) * 2;
    float s = 0.f, ss = 0.f;
    for (int k = lane; k < nchunk; k += 64) {
        s += pp[(int64_t)k * G * 2];
        ss += pp[(int64_t)k * G * 2 + 1];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) {
        const float mean = s / n;
        const float var = fmaxf(ss / n - mean * mean, 0.f);
        stats[i * 2] = mean;
        stats[i * 2 + 1] = rsqrtf(var + eps);
    }
}

constexpr int GN_APPLY_CHUNK = 256;  // pixels per workgroup of the apply pass

template <int DT, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(GnSrc x, const float* stats, const uint8_t* gamma,
                                                       const uint8_t* beta, uint8_t* out, int HW, int C, int G) {
    __shared__ float lm[64], lr[64];
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int cg = C / G, vpr = C >> 3;
    if (tid < G) {
        lm[tid] = stats[((int64_t)b * G + tid) * 2];
        lr[tid] = stats[((int64_t)b * G + tid) * 2 + 1];
    }
    __syncthreads();
    const int p0 = chunk * GN_APPLY_CHUNK, p1 = min(p0 + GN_APPLY_CHUNK, HW);
    if (256 % vpr == 0) {
        // a thread keeps ONE channel vector and walks the chunk's pixels: gamma / beta / the two groups' statistics are loaded once
        // and the loop carries no integer division (the generic loop below spends as many instructions on idx % vpr, c0 / cg as on
        // the normalisation: gn_apply at the 4000-pixel level measured 2.1 TB/s)
        const int vc = tid % vpr, py = tid / vpr, PY = 256 / vpr, c0 = vc * 8;
        float g[8], bt[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(gamma + c0 * 2), g);
        unpack8<DT>(*reinterpret_cast<const uint4*>(beta + c0 * 2), bt);
        const int g0 = c0 / cg, g1 = (c0 + 4) / cg;
        const float m0 = lm[g0], r0 = lr[g0], m1 = lm[g1], r1 = lr[g1];
        const uint8_t* src = gn_vec(x, b, HW, 0, c0);
        const int64_t pstride = (c0 < x.Ca ? x.Ca : x.Cb) * 2;
        uint8_t* dst = out + ((int64_t)b * HW * C + c0) * 2;
        for (int px0 = p0 + py; px0 < p1; px0 += 4 * PY) {  // four pixels per trip: the loads go out together
            uint4 raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = px0 + u * PY;
                if (px < p1) raw[u] = *reinterpret_cast<const uint4*>(src + (int64_t)px * pstride);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = px0 + u * PY;
                if (px >= p1) break;
                float v[8], y[8];
                unpack8<DT>(raw[u], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float mean = e < 4 ? m0 : m1, rstd = e < 4 ? r0 : r1;
                    float t = (v[e] - mean) * rstd * g[e] + bt[e];
                    if (SILU) {
                        t = (float)(typename ET<DT>::elem)t;
                        t = silu_f(t);
                    }
                    y[e] = t;
                }
                *reinterpret_cast<uint4*>(dst + (int64_t)px * C * 2) = pack8<DT>(y);
            }
        }
        return;
    }
    const int nv = (p1 - p0) * vpr;
    uint8_t* ob = out + (((int64_t)b * HW + p0) * C) * 2;
    for (int idx = tid; idx < nv; idx += 256) {
        const int vc = idx % vpr;
        const int c0 = vc * 8;
        float v[8], g[8], bt[8], y[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(gn_vec(x, b, HW, p0 + idx / vpr, c0)), v);
        unpack8<DT>(*reinterpret_cast<const uint4*>(gamma + c0 * 2), g);
        unpack8<DT>(*reinterpret_cast<const uint4*>(beta + c0 * 2), bt);
        const int g0 = c0 / cg, g1 = (c0 + 4) / cg;
        const float m0 = lm[g0], r0 = lr[g0], m1 = lm[g1], r1 = lr[g1];
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
        *reinterpret_cast<uint4*>(ob + (int64_t)idx * 16) = pack8<DT>(y);
    }
}

// One-pass GroupNorm for the low-resolution levels (HW <= 1024 pixels per sample): workgroup = (4 groups, sample).  The
// two-pass schedule above launches only HW/64 x B workgroups there (32 at the 64-pixel level) and pays two launches;
// here a workgroup reads the [HW][4 groups] slab of its sample ONCE into registers (thread (vc, py) keeps the 16-byte
// vectors of pixels py, py+PY, ...), reduces it in a fixed order through LDS, and normalises from the registers.
constexpr int GN1_GPB = 4;  // groups per workgroup of the default form; GPB = 8 / 16: the full-line form for 8- / 4-channel groups (below)

template <int DT, bool SILU, int NTH, int MAXP, int GPB = GN1_GPB>
__global__ __launch_bounds__(NTH) void gn_onepass_kernel(GnSrc x, const uint8_t* gamma, const uint8_t* beta,
                                                         uint8_t* out, int HW, int C, int G, float eps) {
    constexpr int GN1_GPB = GPB;  // (shadows the default)
    __shared__ float lh[NTH][4];   // per thread: (sum, sumsq) of the low and of the high 4 channels of its vector
    __shared__ float lcol[2][48];  // per 4-channel column of the slab (4 groups x cg <= 40 channels / 4)
    __shared__ float lm[GN1_GPB], lr[GN1_GPB];
    const int b = blockIdx.y, tid = threadIdx.x;
    const int cg = C / G, wc = GN1_GPB * cg, vps = wc >> 3;  // cg % 4 == 0 -> a 4-channel half vector lies in one group
    const int PY = NTH / vps;
    const int vc = tid % vps, py = tid / vps;
    const bool active = py < PY;
    const int cbase = blockIdx.x * wc + vc * 8;
    const uint8_t* xb = gn_vec(x, b, HW, 0, cbase);  // (a thread's vector lies in one source)
    const int64_t pstride = (cbase < x.Ca ? x.Ca : x.Cb) * 2;
    uint4 keep[MAXP];
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int px = py + i * PY;
        if (active && px < HW) {
            keep[i] = *reinterpret_cast<const uint4*>(xb + (int64_t)px * pstride);
            float v[8];
            unpack8<DT>(keep[i], v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0 += v[e];
                q0 += v[e] * v[e];
                s1 += v[e + 4];
                q1 += v[e + 4] * v[e + 4];
            }
        }
    }
    lh[tid][0] = s0;
    lh[tid][1] = q0;
    lh[tid][2] = s1;
    lh[tid][3] = q1;
    __syncthreads();
    if (tid < 2 * vps) {  // fixed-order sum over the pixel lanes of one 4-channel column
        const int cv = tid >> 1, h = tid & 1;
        float a = 0.f, q = 0.f;
        for (int k = 0; k < PY; ++k) {
            a += lh[k * vps + cv][2 * h];
            q += lh[k * vps + cv][2 * h + 1];
        }
        lcol[0][tid] = a;
        lcol[1][tid] = q;
    }
    __syncthreads();
    if (tid < GN1_GPB) {
        const int ncol = cg >> 2;
        float a = 0.f, q = 0.f;
        for (int k = tid * ncol; k < (tid + 1) * ncol; ++k) {
            a += lcol[0][k];
            q += lcol[1][k];
        }
        const float n = (float)HW * (float)cg;
        const float mean = a / n;
        const float var = fmaxf(q / n - mean * mean, 0.f);
        lm[tid] = mean;
        lr[tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    if (!active) return;
    float g[8], bt[8];
    unpack8<DT>(*reinterpret_cast<const uint4*>(gamma + cbase * 2), g);
    unpack8<DT>(*reinterpret_cast<const uint4*>(beta + cbase * 2), bt);
    const int g0 = (vc * 8) / cg, g1 = (vc * 8 + 4) / cg;
    const float m0 = lm[g0], r0 = lr[g0], m1 = lm[g1], r1 = lr[g1];
    uint8_t* ob = out + ((int64_t)b * HW * C + cbase) * 2;
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int px = py + i * PY;
        if (px < HW) {
            float v[8], y[8];
            unpack8<DT>(keep[i], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float mean = e < 4 ? m0 : m1, rstd = e < 4 ? r0 : r1;
                float t = (v[e] - mean) * rstd * g[e] + bt[e];
                if (SILU) {
                    t = (float)(typename ET<DT>::elem)t;  // as gn_apply_kernel: storage rounding before SiLU
                    t = silu_f(t);
                }
                y[e] = t;
            }
            *reinterpret_cast<uint4*>(ob + (int64_t)px * C * 2) = pack8<DT>(y);
        }
    }
}

template <int DT, bool SILU>
bool gn_onepass_launch(const GnSrc& x, const void* gamma, const void* beta, void* out, int B, int HW, int C, int G, float eps,
                       hipStream_t s) {
    constexpr int max_hw = 1024;
    const int cg = C / G;
    // full-line form (round 6): with 8- (4-) channel groups the default slab of 4 groups is 64 (32) bytes per pixel -- every 128-byte line is
    // fetched by two (four) workgroups at different times; 8 (16) groups per workgroup make the slab one whole line per pixel, 512 threads keep it
    if (HW <= max_hw && (cg == 8 || cg == 4) && G % (64 / cg) == 0 && x.Cb == 0) {
        dim3 grid(G / (64 / cg), B);
        if (cg == 8)
            hipLaunchKernelGGL((gn_onepass_kernel<DT, SILU, 512, 16, 8>), grid, dim3(512), 0, s, x, (const uint8_t*)gamma, (const uint8_t*)beta,
                               (uint8_t*)out, HW, C, G, eps);
        else
            hipLaunchKernelGGL((gn_onepass_kernel<DT, SILU, 512, 16, 16>), grid, dim3(512), 0, s, x, (const uint8_t*)gamma, (const uint8_t*)beta,
                               (uint8_t*)out, HW, C, G, eps);
        return true;
    }
    if (HW > max_hw || G % GN1_GPB != 0 || cg % 4 != 0 || cg > 40) return false;
    const int vps = GN1_GPB * cg / 8;
    dim3 grid(G / GN1_GPB, B);
    if ((HW + 256 / vps - 1) / (256 / vps) <= 24) {
        hipLaunchKernelGGL((gn_onepass_kernel<DT, SILU, 256, 24>), grid, dim3(256), 0, s, x, (const uint8_t*)gamma,
                           (const uint8_t*)beta, (uint8_t*)out, HW, C, G, eps);
        return true;
    }
    if ((HW + 1024 / vps - 1) / (1024 / vps) <= 12) {
        hipLaunchKernelGGL((gn_onepass_kernel<DT, SILU, 1024, 12>), grid, dim3(1024), 0, s, x,
                           (const uint8_t*)gamma, (const uint8_t*)beta, (uint8_t*)out, HW, C, G, eps);
        return true;
    }
    return false;
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

template <int DT> int gn_launch(const GnSrc& x, const void* gamma, const void* beta, void* out, float* ws, int B, int HW, int C,
                                int G, float eps, int silu, hipStream_t s) {
    if (silu ? gn_onepass_launch<DT, true>(x, gamma, beta, out, B, HW, C, G, eps, s)
             : gn_onepass_launch<DT, false>(x, gamma, beta, out, B, HW, C, G, eps, s))
        return apad_check_launch("apad_groupnorm(one pass)");
    const int nchunk = (HW + GN_CHUNK - 1) / GN_CHUNK;
    hipLaunchKernelGGL((gn_partial_kernel<DT>), dim3(nchunk, B), dim3(256), 0, s, x, ws, HW, C, G, nchunk);
    int rc = apad_check_launch("apad_groupnorm(stats)");
    if (rc) return rc;
    float* stats = ws + (int64_t)B * nchunk * G * 2;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * G + 3) / 4), dim3(256), 0, s, ws, stats, B * G, G, nchunk, (float)HW * (float)(C / G), eps);
    rc = apad_check_launch("apad_groupnorm(finalize)");
    if (rc) return rc;
    const int achunk = (HW + GN_APPLY_CHUNK - 1) / GN_APPLY_CHUNK;
    if (silu)
        hipLaunchKernelGGL((gn_apply_kernel<DT, true>), dim3(achunk, B), dim3(256), 0, s, x, stats,
                           (const uint8_t*)gamma, (const uint8_t*)beta, (uint8_t*)out, HW, C, G);
    else
        hipLaunchKernelGGL((gn_apply_kernel<DT, false>), dim3(achunk, B), dim3(256), 0, s, x, stats,
                           (const uint8_t*)gamma, (const uint8_t*)beta, (uint8_t*)out, HW, C, G);
    return apad_check_launch("apad_groupnorm(apply)");
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int apad_layernorm(const void* x, const void* gamma, const void* beta, void* out, int64_t M, int32_t C, int64_t ldx,
                              int64_t ldo, float eps, int32_t dtype, void* stream) {
    APAD_CHECK(x && gamma && beta && out, "apad_layernorm: null operand");
    if (dtype == APAD_F32) return apad_f32_layernorm(x, gamma, beta, out, M, C, ldx, ldo, eps, (hipStream_t)stream);
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_layernorm: dtype %d not supported", dtype);
    APAD_CHECK(M > 0 && C > 0 && C % 8 == 0 && C <= 2048, "apad_layernorm: need M>0, C%%8==0, C<=2048 (M=%lld C=%d)", (long long)M, C);
    APAD_CHECK(ldx % 8 == 0 && ldo % 8 == 0 && al16(x) && al16(out) && al16(gamma) && al16(beta),
               "apad_layernorm: rows must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    return dtype == APAD_BF16 ? ln_launch<APAD_BF16>(x, gamma, beta, out, M, C, ldx, ldo, eps, s)
                              : ln_launch<APAD_F16>(x, gamma, beta, out, M, C, ldx, ldo, eps, s);
}

extern "C" int64_t apad_groupnorm_workspace_bytes(int32_t B, int32_t HW, int32_t G) {
    return ((int64_t)B * ((HW + GN_CHUNK - 1) / GN_CHUNK) * G * 2 + (int64_t)B * G * 2) * sizeof(float);  // chunk partials + (mean, rstd)
}

extern "C" int apad_groupnorm2(const void* xa, const void* xb, const void* gamma, const void* beta, void* out, void* workspace,
                               int32_t B, int32_t Ba, int32_t Bb, int32_t HW, int32_t Ca, int32_t Cb, int32_t G, float eps, int32_t silu,
                               int32_t dtype, void* stream) {
    APAD_CHECK(xa && gamma && beta && out && workspace, "apad_groupnorm: null operand");
    const int C = Ca + (xb ? Cb : 0);
    if (dtype == APAD_F32) {
        APAD_CHECK(xb == nullptr && Ba == B, "apad_groupnorm2: the fp32 mode takes one source");
        return apad_f32_groupnorm(xa, gamma, beta, out, B, HW, C, G, eps, silu, (hipStream_t)stream);
    }
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_groupnorm: dtype %d not supported", dtype);
    APAD_CHECK(B > 0 && HW > 0 && G > 0 && G <= 64 && C % G == 0 && (C / G) % 4 == 0 && C % 8 == 0 && C <= 1280,
               "apad_groupnorm: need G<=64, C%%G==0, (C/G)%%4==0, C%%8==0, C<=1280 (B=%d HW=%d C=%d G=%d)", B, HW, C, G);
    APAD_CHECK(Ca > 0 && Ca % 8 == 0 && Ba > 0 && B % Ba == 0 && (xb == nullptr || (Cb > 0 && Cb % 8 == 0 && Bb > 0 && B % Bb == 0)),
               "apad_groupnorm2: need Ca%%8==0, Cb%%8==0 and source batches that divide B (B=%d Ba=%d Bb=%d Ca=%d Cb=%d)", B, Ba, Bb, Ca, Cb);
    APAD_CHECK(al16(xa) && al16(xb) && al16(out) && al16(gamma) && al16(beta), "apad_groupnorm: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    GnSrc x;
    x.a = (const uint8_t*)xa; x.b = (const uint8_t*)xb; x.Ca = Ca; x.Cb = xb ? Cb : 0; x.Ba = Ba; x.Bb = xb ? Bb : 1;
    return dtype == APAD_BF16 ? gn_launch<APAD_BF16>(x, gamma, beta, out, (float*)workspace, B, HW, C, G, eps, silu, s)
                              : gn_launch<APAD_F16>(x, gamma, beta, out, (float*)workspace, B, HW, C, G, eps, silu, s);
}

extern "C" int apad_groupnorm(const void* x, const void* gamma, const void* beta, void* out, void* workspace, int32_t B,
                              int32_t HW, int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype, void* stream) {
    return apad_groupnorm2(x, nullptr, gamma, beta, out, workspace, B, B, B, HW, C, 0, G, eps, silu, dtype, stream);
}
