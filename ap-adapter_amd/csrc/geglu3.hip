// apad_layernorm_geglu_packed (round 5): LayerNorm + the GEGLU projection of a feed-forward at C = 384 (the 252-token level),
//     H = value * gelu(gate),   [value | gate] = W1 . LayerNorm(x) + b1        (diffusers GEGLU behind norm3)
// as a 64-TOKEN REGISTER BLOCK per wave -- mlp3.hip's loop without its second GEMM (at C = 384 the output accumulators of a whole feed-forward do not
// fit beside the x panels: 192 + 384 registers; the 4C -> C product stays on the big-tile GEMM).
//   * a wave owns 64 tokens (two 32-token panels): x, normalised once, lives in 192 AGPRs (the B operands of the inline-asm MFMAs), the first-GEMM
//     accumulators in VGPRs, where the GEGLU arithmetic reads them; one wave per SIMD;
//   * the 1536 hidden units are split over FOUR workgroups per 256-token tile (16 128 rows = 63 tiles -> 252 workgroups, one per CU; the four of a
//     tile share an XCD, so its rows are fetched into one L2); a workgroup walks its 24 chunks of 16 units;
//   * the weights come packed (apad_geglu_pack): per chunk 24 fragments of 1 KB (16 value rows + their 16 gate rows x a k-step of 16), lane-linear,
//     L2 -> LDS by `buffer_load ... lds` into a ring of six 24 KB slots, five chunks ahead, counted vmcnt, one raw s_barrier per chunk; every
//     fragment (one conflict-free ds_read_b128) feeds two MFMAs;
//   * per chunk a wave issues 48 MFMAs with the GEGLU of the previous chunk placed between them in phases; the activated chunk leaves as one
//     16-byte store per lane and panel (the half-waves exchange 8-byte pieces first).
// Numerics: b1 as the accumulators' initial value, the same MFMA k-order, gelu_erf_2's arithmetic, the fp32 product rounded once -- bit-equal to
// rpgemm_kernel<..., GEGLU> (the form smaller launches take), so a row's result does not depend on its batch.
#include <stdlib.h>
#include "mlp3_shared.h"

#ifndef G3_ABL
#define G3_ABL 0  // timing ablations (results are wrong): 1 no GELU arithmetic, 2 no MFMAs, 8 no DMA in the loop, 16 no barrier, 32 no output stores, 64 no fragment reads, 128 two iterations only
#endif

namespace {

constexpr int G3_C = 384, G3_KS = G3_C / 16, G3_HID = 4 * G3_C, G3_PARTS = 4, G3_NCH = G3_HID / 16 / G3_PARTS;  // 24 k-steps; 24 chunks per part
constexpr int G3_NW = 8;                           // waves per workgroup: 32 tokens each, two per SIMD
constexpr int G3_STAGE = G3_KS * 1024;             // 24 576: the 24 fragments of one chunk
constexpr int G3_NS = 6;
constexpr int G3_RING = G3_NS * G3_STAGE;          // 147 456
constexpr int G3_B1_BYTES = G3_NCH * 2 * 16 * 4;   // this part's b1, fp32 [chunk][half][16] in C-layout register order
constexpr int G3_TROWS = 32 * 400;                 // a wave's transposition region (32 rows of 384 bytes + 16 of padding)
constexpr int G3_B1_OFF = 2 * G3_STAGE + G3_NW * G3_TROWS;  // the transposition regions start at slot 2 and run past the ring's end
constexpr int G3_LDS = G3_B1_OFF + G3_B1_BYTES;
static_assert(G3_B1_OFF >= G3_RING && G3_LDS <= 160 * 1024, "LDS layout");

struct G3P {
    const uint8_t* x;
    const uint8_t* gamma;
    const uint8_t* beta;
    const uint8_t* wpk;   // [4 parts][24 chunks][24 fragments][64 lanes][8]
    const float* b1p;     // [4 parts][24 chunks][2][16]
    uint8_t* out;         // [M][4C]
    int64_t M;
    int32_t ntile;
    float eps;
};

// the B operand (x) comes from AGPRs: 192 of them hold the two panels for the whole kernel
template <int DT> struct G3Asm;
template <> struct G3Asm<APAD_BF16> {
    template <typename V8> static __device__ __forceinline__ void first(f32x16& d, const V8& a, const V8& b, const f32x16& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
    template <typename V8> static __device__ __forceinline__ void acc(f32x16& d, const V8& a, const V8& b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
};
template <> struct G3Asm<APAD_F16> {
    template <typename V8> static __device__ __forceinline__ void first(f32x16& d, const V8& a, const V8& b, const f32x16& c) {
        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
    }
    template <typename V8> static __device__ __forceinline__ void acc(f32x16& d, const V8& a, const V8& b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
    }
};

// probe build (tools/ab_build.sh <tag> geglu3.hip -DG3_TRACE=<wave>; tools/g3_trace.py): s_memtime at the phase boundaries of one wave of every workgroup
#ifdef G3_TRACE
__device__ unsigned long long g3_trace_buf[1024][40];
#define G3_STAMP(i_)                                                                                                 \
    if (lane == 0 && wave == (G3_TRACE) && blockIdx.x < 1024) {                                                      \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        g3_trace_buf[blockIdx.x][i_] = __builtin_amdgcn_s_memtime();                                                 \
        if ((i_) == 0) g3_trace_buf[blockIdx.x][38] = wall_clock64();                                                \
        if ((i_) == 37) g3_trace_buf[blockIdx.x][39] = wall_clock64();                                               \
    }
#else
#define G3_STAMP(i_)
#endif

template <int OFF> __device__ __forceinline__ void g3_write(uint32_t a, const u32x4& d) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a), "v"(d), "n"(OFF) : "memory");
}

#define G3_RD2(k_, f_) do { if (!(G3_ABL & 64)) m3_read2<k_>(f_, fa); } while (0)

template <int DT, bool LN>
__global__ __launch_bounds__(512, 1) void geglu3_kernel(G3P p) {
    using E = ET<DT>;
    using V8 = typename E::v8;
    using EL = typename E::elem;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // the four parts of a tile on one XCD (workgroup id w runs on XCD w % 8: observed, speed only)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int mtile = (seq >> 2) * 8 + xcd, part = seq & 3;
    if (mtile >= p.ntile) return;
    const int64_t mw0 = ((int64_t)mtile * G3_NW + wave) * 32;
    G3_STAMP(0);

    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.wpk), 0, G3_PARTS * G3_NCH * G3_STAGE, 0x00020000);
    const uint32_t dvoff = (uint32_t)(lane * 16);
    const int part_off = part * G3_NCH * G3_STAGE;
    auto dma = [&](int stage, int slot, int q) __attribute__((always_inline)) {
        if (G3_ABL & 8) return;
        // (the immediate offset moves the memory address AND the LDS address: the wave's three 1 KB pieces of a stage share one M0 / scalar offset)
        const m3_lds_ptr lp = (m3_lds_ptr)(smem + slot * G3_STAGE + wave * 3072);
        const int so = part_off + stage * G3_STAGE + wave * 3072;
        switch (q) {
            case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lp, 16, dvoff, so, 0, 0); break;
            case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lp, 16, dvoff, so, 1024, 0); break;
            default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lp, 16, dvoff, so, 2048, 0); break;
        }
    };

    // ---- the wave's 32 rows of x: 24 COALESCED 16-byte loads per lane (a load instruction covers 1 KB in runs of 384 bytes; the fragment layout --
    // lane = token -- asked of global memory directly touches 32 cache lines per instruction and measured 225 cycles of address processing each),
    // transposed to the MFMA fragment layout through a wave-private LDS region (rows padded to 400 bytes: conflict-free both ways), half a row at a time
    u32x4 stg[2][12];
    {
        // (scalar base + 32-bit lane offset form: 64-bit per-lane pointers for the 24 loads overflowed the 128-register budget of the prologue)
        typedef const __attribute__((address_space(1))) uint8_t* g3_gptr;
        typedef const __attribute__((address_space(1))) u32x4* g3_gptr16;
        const int64_t mb = mw0 < p.M ? mw0 : p.M - 1;  // (a wave wholly past the end re-reads the last row; mtile < ntile: M >= 1)
        const uint64_t xa = reinterpret_cast<uint64_t>(p.x) + (uint64_t)mb * (G3_C * 2);
        const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xa), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xa >> 32));  // (unsigned: no sign extension)
        const g3_gptr xb = (g3_gptr)(((uint64_t)xhi << 32) | xlo);
        const int rmax = (int)(p.M - 1 - mb < 31 ? p.M - 1 - mb : 31);
        uint32_t go[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int g = i * 64 + lane, row = g / 24, c = g - row * 24;
            go[i] = (uint32_t)((row < rmax ? row : rmax) * (G3_C * 2) + c * 16);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 12; ++i) stg[h][i] = *(g3_gptr16)(xb + go[i] + h * 384);
    }
    G3_STAMP(1);
    // this part's bias table (3 KB) -> LDS by the DMA as well (waves 0..2, one 1 KB piece each)
    if (wave < 3) {
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.b1p), 0, G3_PARTS * G3_B1_BYTES, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (m3_lds_ptr)(smem + G3_B1_OFF + wave * 1024), 16, dvoff, part * G3_B1_BYTES + wave * 1024, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) dma(s, s, q);  // (stages 0, 1 -> slots 0, 1; slots 2.. hold the transposition until the barrier below)
    G3_STAMP(2);
    V8 xf[G3_KS];
    {
        const uint32_t tw = (uint32_t)(size_t)(m3_lds_ptr)smem + (uint32_t)(2 * G3_STAGE + wave * G3_TROWS);
        uint32_t wa[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int g = i * 64 + lane, row = g / 24, c = g - row * 24;
            wa[i] = tw + (uint32_t)(row * 400 + c * 16);
        }
        const uint32_t ra = tw + (uint32_t)(l31 * 400 + half * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // (behind these loads: the other half's 12, up to 7 DMA pieces -- the compiler's own count for `stg` is what orders this; no explicit wait)
#pragma unroll
            for (int i = 0; i < 12; ++i) g3_write<0>(wa[i], stg[h][i]);
            u32x4 t[12];
            m3_read<0>(t[0], ra); m3_read<32>(t[1], ra); m3_read<64>(t[2], ra); m3_read<96>(t[3], ra);
            m3_read<128>(t[4], ra); m3_read<160>(t[5], ra); m3_read<192>(t[6], ra); m3_read<224>(t[7], ra);
            m3_read<256>(t[8], ra); m3_read<288>(t[9], ra); m3_read<320>(t[10], ra); m3_read<352>(t[11], ra);
            m3_wait_lgkm<0>();
#pragma unroll
            for (int k = 0; k < 12; ++k) xf[h * 12 + k] = __builtin_bit_cast(V8, t[k]);
        }
    }
    if (LN) layernorm_panel<DT, G3_KS, true>(xf, p.gamma, p.beta, p.eps, l31, half);
    G3_STAMP(3);
    // the panel moves to AGPRs HERE, once (left to the allocator, part of it stays in the VGPRs the LayerNorm wrote and is copied in front of every MFMA)
#pragma unroll
    for (int k = 0; k < G3_KS; ++k) asm volatile("" : "+a"(xf[k]));

    f32x16 a0, b0;  // alternate between "being activated" and "being accumulated"
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = b0[r] = 0.f;

    const uint32_t lds0 = (uint32_t)(size_t)(m3_lds_ptr)smem;
    const uint32_t fbase = lds0 + (uint32_t)(lane * 16);
    const uint32_t tbase = lds0 + (uint32_t)(G3_B1_OFF + half * 64);
    // this lane's output row pointer: token l31 of the wave's panel, the 16-byte piece of its half-wave
    const int64_t m0 = mw0 + l31;
    uint8_t* const orow = p.out + ((m0 < p.M ? m0 : 0) * G3_HID + part * (G3_NCH * 16) + 8 * half) * 2;
    const bool ok = m0 < p.M;
    G3_STAMP(4);
    __syncthreads();  // the bias table is in LDS, every wave is done with the transposition region
#pragma unroll
    for (int s = 2; s < G3_NS - 1; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) dma(s, s, q);
    G3_STAMP(5);

    // the activated chunk c (C layout: units 4 half + {0..3}, 8 + 4 half + {0..3}) -> 8 consecutive units per lane -> one 16-byte store
    auto store_h = [&](const V8& hn, int c) __attribute__((always_inline)) {
        const uint4 u = as_u4<DT>(hn);
        const auto s0 = __builtin_amdgcn_permlane32_swap(u.x, u.z, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(u.y, u.w, false, false);
        if (ok && !(G3_ABL & 32)) *reinterpret_cast<uint4*>(orow + c * 32) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    };

    // iteration i: the 24 MFMAs of chunk i into `anxt` with the GEGLU of chunk i - 1 (`acur`) one phase behind each of the first 16, then the wave's
    // three DMA pieces of chunk i + 5 and the store of chunk i - 1.  Two waves share a SIMD: one wave's phase covers the other's MFMA.
    auto iteration = [&](int i, int slot, f32x16& acur, f32x16& anxt) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // stage i has landed for this wave's pieces (stores only make the count stricter)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(G3_ABL & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t fa = fbase + (uint32_t)(slot * G3_STAGE);
        const uint32_t ta = tbase + (uint32_t)(i * 128);
        const int nstage = i + G3_NS - 1 < G3_NCH ? i + G3_NS - 1 : G3_NCH - 1;  // (past the end: a dummy re-load keeps the vmcnt arithmetic uniform)
        const int nslot = slot == 0 ? G3_NS - 1 : slot - 1;

        u32x4 bq[4], f0[2], f1[2], f2[2];  // fragments two steps ahead (eight waves share the LDS pipe: one step ahead left every step waiting on it)
        if (G3_ABL & 64) f0[0] = f0[1] = f1[0] = f1[1] = f2[0] = f2[1] = u32x4{0u, 0u, 0u, 0u};
        V8 hn;
        M3Geglu<DT> gg;
        m3_read<0>(bq[0], ta);
        m3_read<16>(bq[1], ta);
        m3_read<32>(bq[2], ta);
        m3_read<48>(bq[3], ta);
        G3_RD2(0, f0);
        G3_RD2(2, f1);
        __builtin_amdgcn_sched_barrier(0);

        // MFMA slot m (k-step m) and what follows it: the 16 GEGLU phases of chunk i - 1 behind two of every three MFMAs, the wave's three DMA pieces and
        // nothing behind the others (vector issue is the shared resource of the SIMD's two waves: ~5 cycles per instruction whichever wave it comes
        // from -- tools/ubench/pair.hip -- so the phases are spread over the whole chunk instead of crowding its first 16 MFMAs)
        auto after = [&](int m) __attribute__((always_inline)) {
            const int g = m / 3, o = m % 3;           // group of three MFMAs: phases 2 g, 2 g + 1 behind the first two
            if (o < 2) {
                const int ph = 2 * g + o, r = 2 * (ph / 4);
                if (!(G3_ABL & 1)) {
                    if ((ph & 3) == 0) gg.ph1(acur[8 + r], acur[9 + r]);
                    else if ((ph & 3) == 1) gg.ph2();
                    else if ((ph & 3) == 2) gg.ph3();
                    else gg.template ph4<V8, EL>(acur[r], acur[r + 1], hn, r);
                }
            } else if (g >= 1 && g <= 3) {
                dma(nstage, nslot, g - 1);
            }
        };
        auto mf = [&](int m, const V8& w) __attribute__((always_inline)) {
            if (!(G3_ABL & 2)) G3Asm<DT>::acc(anxt, w, xf[m]);
            M3_PIN();
            after(m);
            M3_PIN();
        };
        auto step = [&](const u32x4 (&f)[2], int ks) __attribute__((always_inline)) {
            mf(ks, __builtin_bit_cast(V8, f[0]));
            mf(ks + 1, __builtin_bit_cast(V8, f[1]));
        };

        // ---- step 0: b1 (C-layout register order) is the C operand of the first MFMA ----
        G3_RD2(4, f2);
        m3_wait_lgkm<4>();
        {
            f32x16 bias;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[qd * 4 + e] = __uint_as_float(bq[qd][e]);
            if (!(G3_ABL & 2)) {
                G3Asm<DT>::first(anxt, __builtin_bit_cast(V8, f0[0]), xf[0], bias);
                asm volatile("s_nop 7\n\ts_nop 6" ::: "memory");  // (the b1 registers may be recycled right here: 13 wait states behind an MFMA that reads them as C)
            } else {
                anxt = bias;
            }
            M3_PIN();
            after(0);
            M3_PIN();
            mf(1, __builtin_bit_cast(V8, f0[1]));
        }
        G3_RD2(6, f0);
        m3_wait_lgkm<4>();
        step(f1, 2);
        G3_RD2(8, f1);
        m3_wait_lgkm<4>();
        step(f2, 4);
        G3_RD2(10, f2);
        m3_wait_lgkm<4>();
        step(f0, 6);
        G3_RD2(12, f0);
        m3_wait_lgkm<4>();
        step(f1, 8);
        G3_RD2(14, f1);
        m3_wait_lgkm<4>();
        step(f2, 10);
        G3_RD2(16, f2);
        m3_wait_lgkm<4>();
        step(f0, 12);
        G3_RD2(18, f0);
        m3_wait_lgkm<4>();
        step(f1, 14);
        G3_RD2(20, f1);
        m3_wait_lgkm<4>();
        step(f2, 16);
        G3_RD2(22, f2);
        m3_wait_lgkm<4>();
        step(f0, 18);
        m3_wait_lgkm<2>();
        step(f1, 20);
        m3_wait_lgkm<0>();
        step(f2, 22);
        if (i >= 1) store_h(hn, i - 1);
    };

    int slot = 0;
#pragma unroll 1
    for (int i = 0; i < ((G3_ABL & 128) ? 2 : G3_NCH); i += 2) {
        G3_STAMP(6 + i);
        iteration(i, slot, a0, b0);
        slot = slot == G3_NS - 1 ? 0 : slot + 1;
        G3_STAMP(7 + i);
        iteration(i + 1, slot, b0, a0);
        slot = slot == G3_NS - 1 ? 0 : slot + 1;
    }
    static_assert(G3_NCH % 2 == 0, "the loop body is two iterations");
    G3_STAMP(36);
    // ---- the last chunk's GEGLU (a0 after an even number of iterations) ----
    {
        V8 hn;
        M3Geglu<DT> gg;
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // (the accumulators were written by the MFMAs right above)
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            gg.ph1(a0[8 + r], a0[9 + r]);
            gg.ph2();
            gg.ph3();
            gg.template ph4<V8, EL>(a0[r], a0[r + 1], hn, r);
        }
        store_h(hn, G3_NCH - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the tail's dummy re-loads must land before the workgroup's LDS is released)
    G3_STAMP(37);
}

// ---- apad_geglu_pack: W1 [8C][C], b1 [8C] -> [4 parts][24 chunks][24 k-steps][64 lanes][8] + the fp32 bias table [4][24][2][16] ----
template <int DT>
__global__ void geglu3_pack_kernel(const uint8_t* w1, const uint8_t* b1, uint8_t* wpk, float* b1p) {
    using elem = typename ET<DT>::elem;
    const int64_t total = (int64_t)G3_PARTS * G3_NCH * (G3_STAGE / 2);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(e / (G3_STAGE / 2)), r = (int)(e % (G3_STAGE / 2));  // ch = part * 24 + chunk: units 16 ch .. 16 ch + 15
        const int ks = r / 512, rem = r % 512, lane = rem / 8, j = rem % 8, hf = lane >> 5, l31 = lane & 31;
        const int row = l31 < 16 ? ch * 16 + l31 : G3_HID + ch * 16 + (l31 - 16);
        const elem v = reinterpret_cast<const elem*>(w1)[(int64_t)row * G3_C + ks * 16 + hf * 8 + j];
        reinterpret_cast<elem*>(wpk)[e] = l31 < 16 ? (elem)((float)v * m3_value_scale<DT>()) : v;  // (the value rows carry GEGLU's 1/2: M3GegluT<PRE>)
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < G3_PARTS * G3_NCH * 32; t += gridDim.x * blockDim.x) {
        const int ch = t / 32, hf = (t % 32) / 16, r = t % 16, u = ((r & 7) & 3) + 8 * ((r & 7) >> 2) + 4 * hf;
        b1p[t] = b1 != nullptr ? (float)reinterpret_cast<const elem*>(b1)[(r < 8 ? 0 : G3_HID) + ch * 16 + u] * (r < 8 ? m3_value_scale<DT>() : 1.0f) : 0.f;
    }
}

inline bool g3_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int DT, bool LN> int geglu3_launch(const G3P& p, hipStream_t s) {
    auto kern = geglu3_kernel<DT, LN>;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), G3_LDS, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(((p.ntile + 7) / 8) * 8 * G3_PARTS)), dim3(512), G3_LDS, s, p);
    return apad_check_launch("apad_layernorm_geglu_packed");
}

}  // namespace

#ifdef G3_TRACE
extern "C" int apad_g3_trace_read(void* dst, int bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g3_trace_buf), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t apad_geglu_packed_bytes(int32_t C) { return C == G3_C ? (int64_t)G3_PARTS * G3_NCH * G3_STAGE : -1; }
extern "C" int64_t apad_geglu_packed_bias_floats(int32_t C) { return C == G3_C ? (int64_t)G3_PARTS * G3_NCH * 32 : -1; }

extern "C" int apad_geglu_pack(const void* w1, const void* b1, void* w_packed, float* b1_packed, int32_t C, int32_t dtype, void* stream) {
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_geglu_pack: dtype %d not supported", dtype);
    APAD_CHECK(w1 && w_packed && b1_packed, "apad_geglu_pack: null operand");
    if (C != G3_C) {
        apad_set_error("apad_geglu_pack: C=%d outside the kernel envelope (384)", C);
        return -3;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_BF16)
        hipLaunchKernelGGL(geglu3_pack_kernel<APAD_BF16>, dim3(512), dim3(256), 0, s, (const uint8_t*)w1, (const uint8_t*)b1, (uint8_t*)w_packed, b1_packed);
    else
        hipLaunchKernelGGL(geglu3_pack_kernel<APAD_F16>, dim3(512), dim3(256), 0, s, (const uint8_t*)w1, (const uint8_t*)b1, (uint8_t*)w_packed, b1_packed);
    return apad_check_launch("apad_geglu_pack");
}

extern "C" int apad_layernorm_geglu_packed(const void* x, const void* ln_gamma, const void* ln_beta, const void* w_packed, const float* b1_packed, void* out,
                                           int64_t M, int32_t C, float ln_eps, int32_t dtype, void* stream) {
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_layernorm_geglu_packed: dtype %d not supported", dtype);
    APAD_CHECK(x && w_packed && b1_packed && out && M > 0, "apad_layernorm_geglu_packed: null operand / empty problem");
    APAD_CHECK((ln_gamma == nullptr) == (ln_beta == nullptr), "apad_layernorm_geglu_packed: LayerNorm needs gamma and beta");
    APAD_CHECK(g3_al16(x) && g3_al16(w_packed) && g3_al16(b1_packed) && g3_al16(out) && g3_al16(ln_gamma) && g3_al16(ln_beta),
               "apad_layernorm_geglu_packed: pointers must be 16-byte aligned");
    if (C != G3_C) {
        apad_set_error("apad_layernorm_geglu_packed: C=%d outside the kernel envelope (384)", C);
        return -3;
    }
    G3P p;
    p.x = (const uint8_t*)x; p.gamma = (const uint8_t*)ln_gamma; p.beta = (const uint8_t*)ln_beta; p.wpk = (const uint8_t*)w_packed; p.b1p = b1_packed;
    p.out = (uint8_t*)out; p.M = M; p.ntile = (int32_t)((M + 255) / 256); p.eps = ln_eps;
    hipStream_t s = (hipStream_t)stream;
    const bool ln = ln_gamma != nullptr;
    if (dtype == APAD_BF16) return ln ? geglu3_launch<APAD_BF16, true>(p, s) : geglu3_launch<APAD_BF16, false>(p, s);
    return ln ? geglu3_launch<APAD_F16, true>(p, s) : geglu3_launch<APAD_F16, false>(p, s);
}
