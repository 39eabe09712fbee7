// apad_geglu_mlp_packed (round 5): the fused feed-forward of mlp.hip -- out = x + W2 . (value * gelu(gate)) + b2, [value | gate] = W1 . LayerNorm(x) + b1,
// diffusers FeedForward / GEGLU -- as a 64-TOKEN REGISTER BLOCK per wave.
//
// What bounded mlp2_kernel (32 tokens per wave, two waves per SIMD): every 32x32x16 MFMA took one new 1 KB weight fragment from LDS, the weights were
// staged global -> registers -> ds_write (a third of the LDS pipe's time) behind a __syncthreads per 32 hidden units, and the W2 fragments were two
// 8-byte reads each.  Here:
//   * a wave owns 64 tokens (two 32-token panels, x in 128 VGPRs, normalised once) and ALL 256 output columns (2 x 8 accumulator tiles = 256 AGPRs): one
//     wave per SIMD, 512 registers; every weight fragment read from LDS feeds TWO MFMAs (one per panel);
//   * the weights are packed ONCE on the host side of the C ABI (apad_mlp_pack) into the exact stream the loop consumes -- per 16-unit chunk 16 W1
//     fragments + 8 W2 fragments of 1 KB, lane-linear, the C-layout -> B-operand permutation of the hidden units baked into the W2 fragments -- and go
//     L2 -> LDS by `buffer_load ... lds` into a ring of six 24 KB slots, requested five chunks ahead, waited for with a counted vmcnt, one raw s_barrier
//     per chunk: no staging registers, no ds_write, no __syncthreads, every fragment read one conflict-free ds_read_b128;
//   * per chunk a wave issues 48 MFMAs (gemm2 of chunk i-2, then gemm1 of chunk i) with the GEGLU arithmetic of chunk i-1 dealt between them -- MFMA and
//     VALU overlap only inside ONE wave's instruction stream on this part (NOTES §4b), which is exactly what a one-wave-per-SIMD kernel offers;
//   * b1 enters as the C operand of the first gemm1 MFMA (no per-element add).
// Workgroup = 4 waves = 256 tokens (250 workgroups at the 1000-token level's 64 000 rows: one per CU, one round).  Numerics: the same MFMA k-order per
// 16-unit chunk, the same GELU (gelu_erf_2's operation order) and the same rounding points as mlp_kernel / mlp2_kernel.
#include <stdlib.h>
#include "mlp3_shared.h"

#ifndef M3_DMA_IMM
#define M3_DMA_IMM 1  // 1: a wave's pieces of a stage share one LDS base / scalar offset, the piece picked by the instruction's immediate offset
#endif
#ifndef M3_DMA_BARE
#define M3_DMA_BARE 1  // 1: the DMA instructions sit between the MFMAs of the gemm1 steps that carry no GEGLU arithmetic
#endif
#ifndef M3_ABL
#define M3_ABL 0  // timing ablations (results are wrong): 1 no GELU arithmetic, 2 no gemm1 MFMAs, 4 no gemm2 MFMAs, 8 no DMA in the loop, 16 no barrier
#endif

namespace {

constexpr int M3_C = 256, M3_HID = 1024, M3_NCH = M3_HID / 16;  // 64 chunks of 16 hidden units
constexpr int M3_NIT = M3_NCH + 2;                                // iteration i: gemm1(i) | GEGLU(i - 1) | gemm2(i - 2)
constexpr int M3_STAGE = 24 * 1024;                               // what iteration i reads: 16 W1 fragments of chunk i, 8 W2 fragments of chunk i - 2
constexpr int M3_NS = 6;                                          // ring slots
constexpr int M3_RING = M3_NS * M3_STAGE;                         // 147 456
constexpr int M3_B1_BYTES = M3_NIT * 2 * 16 * 4;                  // fp32 [NIT][half][16]: b1 in C-layout register order
constexpr int M3_LDS = M3_RING + M3_B1_BYTES + M3_C * 4;          // + b2 (fp32) = 156 928
constexpr int M3_OROWB = M3_C * 2 + 16;                           // epilogue tile row stride (bytes)
static_assert(4 * 64 * M3_OROWB <= M3_RING, "the four waves' output tiles fit the dead ring");

struct Mlp3P {
    const uint8_t* x;
    const uint8_t* gamma;
    const uint8_t* beta;
    const uint8_t* wpk;   // packed weight stream, M3_NIT * M3_STAGE bytes
    const float* b1p;     // [M3_NIT][2][16]
    const uint8_t* b2;
    uint8_t* out;
    int64_t M;
    float eps;
};

template <int VM_, bool DMA_, bool G1_, bool GG_> struct M3It {
    static constexpr int VM = VM_;
    static constexpr bool DMA = DMA_, G1 = G1_, GG = GG_;
};

template <int DT, bool LN>
__global__ __launch_bounds__(256, 1) void mlp3_kernel(Mlp3P p) {
    using E = ET<DT>;
    using V8 = typename E::v8;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int64_t mw0 = ((int64_t)blockIdx.x * 4 + wave) * 64;

    // ---- the weight stream: piece q (0..5) of this wave = 1 KB = one MFMA operand fragment; stage s -> ring slot s % 6 ----
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.wpk), 0, M3_NIT * M3_STAGE, 0x00020000);
    const uint32_t dvoff = (uint32_t)(lane * 16);
    auto dma = [&](int stage, int slot, int q) __attribute__((always_inline)) {
        if (M3_ABL & 8) return;
#if M3_DMA_IMM
        // the instruction's immediate offset moves the memory address AND the LDS address: pieces 0..3 of the wave share one M0 / scalar offset,
        // pieces 4, 5 the next (the packed stream and the ring slot have the same piece order)
        const int grp = q >> 2;
        const m3_lds_ptr lp = (m3_lds_ptr)(smem + slot * M3_STAGE + wave * 6144 + grp * 4096);
        const int so = stage * M3_STAGE + wave * 6144 + grp * 4096;
        switch (q & 3) {
            case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lp, 16, dvoff, so, 0, 0); break;
            case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lp, 16, dvoff, so, 1024, 0); break;
            case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lp, 16, dvoff, so, 2048, 0); break;
            default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, lp, 16, dvoff, so, 3072, 0); break;
        }
#else
        const int piece = wave * 6 + q;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (m3_lds_ptr)(smem + slot * M3_STAGE + piece * 1024), 16, dvoff, stage * M3_STAGE + piece * 1024, 0, 0);
#endif
    };

    // ---- x panels -> registers (requested first: HBM latency), biases -> LDS (fp32; ahead of the first DMA: an LDS store the compiler can see is
    //      ordered behind every LDS-DMA in flight), the first five stages requested, LayerNorm ----
    V8 xf0[16], xf1[16];
    load_panel<DT, 16>(xf0, p.x, M3_C, p.M, mw0, l31, half);
    load_panel<DT, 16>(xf1, p.x, M3_C, p.M, mw0 + 32, l31, half);
    float* const lb1 = reinterpret_cast<float*>(smem + M3_RING);
    float* const lb2 = lb1 + M3_B1_BYTES / 4;
    for (int i = tid; i < M3_B1_BYTES / 4; i += 256) lb1[i] = p.b1p[i];
    for (int i = tid; i < M3_C; i += 256) lb2[i] = p.b2 ? ld_elem<DT>(p.b2, i) : 0.f;
#pragma unroll
    for (int s = 0; s < M3_NS - 1; ++s)
#pragma unroll
        for (int q = 0; q < 6; ++q) dma(s, s, q);
    if (LN) {
        layernorm_panel<DT, 16>(xf0, p.gamma, p.beta, p.eps, l31, half);
        layernorm_panel<DT, 16>(xf1, p.gamma, p.beta, p.eps, l31, half);
    }

    f32x16 y0[8], y1[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) y0[ct][r] = y1[ct][r] = 0.f;
    f32x16 a0, a1, b0, b1;  // first-GEMM accumulators of the two panels: (a*, b*) alternate between "being activated" and "being accumulated"
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = b0[r] = b1[r] = 0.f;
    V8 h0a, h1a, h0b, h1b;  // activations handed to gemm2: (h*a, h*b) alternate between "read" and "written"
#pragma unroll
    for (int r = 0; r < 8; ++r) h0a[r] = h1a[r] = h0b[r] = h1b[r] = (typename E::elem)0.f;

    const uint32_t lds0 = (uint32_t)(size_t)(m3_lds_ptr)smem;
    const uint32_t fbase = lds0 + (uint32_t)(lane * 16);
    const uint32_t tbase = lds0 + (uint32_t)(M3_RING + half * 64);
    __syncthreads();  // the bias tables are in LDS (this also drains the prologue's DMA: the compiler waits vmcnt(0) here)

    // one iteration.  acur*: gemm1 result (+ b1) of chunk i - 1 -> activated here; anxt*: gemm1 of chunk i accumulates here;
    // hp*: activations of chunk i - 2 (gemm2's B operand); hn*: activations of chunk i - 1 (written here)
    auto iteration = [&](auto cfg, int i, int slot, f32x16& acur0, f32x16& acur1, f32x16& anxt0, f32x16& anxt1, const V8& hp0, const V8& hp1, V8& hn0, V8& hn1) __attribute__((always_inline)) {
        using CF = decltype(cfg);  // M3It<VM, DMA, G1, GG>: outstanding pieces allowed at the top, DMA issue / gemm1 / GEGLU of this iteration on or off
        // stage i has landed for this wave's pieces (VM younger requests may stay in flight); every wave is past its reads of stage i - 1
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CF::VM) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(M3_ABL & 16)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t fa = fbase + (uint32_t)(slot * M3_STAGE);
        const uint32_t ta = tbase + (uint32_t)(i * 128);
        const int nstage = i + M3_NS - 1 < M3_NIT ? i + M3_NS - 1 : M3_NIT - 1;  // (past the end: a dummy re-load keeps the vmcnt arithmetic uniform)
        const int nslot = slot == 0 ? M3_NS - 1 : slot - 1;                      // the slot stage i - 1 just left

        u32x4 bq[4], fA[2], fB[2];
        m3_read2<16>(fA, fa);  // W2 tiles 0, 1
        __builtin_amdgcn_sched_barrier(0);

        using EL = typename E::elem;
        M3Geglu<DT> gg;
        // one step of gemm2 (chunk i - 2): output-column tiles ct, ct + 1 x two panels, the GEGLU of hidden units r, r + 1 of panel 0 between the MFMAs
        auto step2 = [&](const u32x4 (&f)[2], int ct, int r, int q) __attribute__((always_inline)) {
            const V8 w0 = __builtin_bit_cast(V8, f[0]), w1 = __builtin_bit_cast(V8, f[1]);
            if (CF::DMA && !M3_DMA_BARE) dma(nstage, nslot, q);
            if (!(M3_ABL & 4)) y0[ct] = E::mfma32(w0, hp0, y0[ct]);
            M3_PIN();
            if (CF::GG && !(M3_ABL & 1)) gg.ph1(acur0[8 + r], acur0[9 + r]);
            M3_PIN();
            if (!(M3_ABL & 4)) y1[ct] = E::mfma32(w0, hp1, y1[ct]);
            M3_PIN();
            if (CF::GG && !(M3_ABL & 1)) gg.ph2();
            M3_PIN();
            if (!(M3_ABL & 4)) y0[ct + 1] = E::mfma32(w1, hp0, y0[ct + 1]);
            M3_PIN();
            if (CF::GG && !(M3_ABL & 1)) gg.ph3();
            M3_PIN();
            if (!(M3_ABL & 4)) y1[ct + 1] = E::mfma32(w1, hp1, y1[ct + 1]);
            M3_PIN();
            if (CF::GG && !(M3_ABL & 1)) gg.template ph4<V8, EL>(acur0[r], acur0[r + 1], hn0, r);
            M3_PIN();
        };
        // one step of gemm1 (chunk i): k-steps ks, ks + 1 x two panels; r >= 0: the GEGLU of hidden units r, r + 1 of panel 1 between the MFMAs
        auto step1 = [&](const u32x4 (&f)[2], int ks, int r, int q, int q2) __attribute__((always_inline)) {
            const V8 w0 = __builtin_bit_cast(V8, f[0]), w1 = __builtin_bit_cast(V8, f[1]);
            if (CF::DMA && !M3_DMA_BARE && q >= 0) dma(nstage, nslot, q);
            if (CF::G1 && !(M3_ABL & 2)) M3Asm<DT>::acc(anxt0, w0, xf0[ks]);
            M3_PIN();
            if (CF::DMA && M3_DMA_BARE && q >= 0) dma(nstage, nslot, q);
            if (CF::GG && r >= 0 && !(M3_ABL & 1)) gg.ph1(acur1[8 + r], acur1[9 + r]);
            M3_PIN();
            if (CF::G1 && !(M3_ABL & 2)) M3Asm<DT>::acc(anxt1, w0, xf1[ks]);
            M3_PIN();
            if (CF::GG && r >= 0 && !(M3_ABL & 1)) gg.ph2();
            M3_PIN();
            if (CF::G1 && !(M3_ABL & 2)) M3Asm<DT>::acc(anxt0, w1, xf0[ks + 1]);
            M3_PIN();
            if (CF::DMA && M3_DMA_BARE && q2 >= 0) dma(nstage, nslot, q2);
            if (CF::GG && r >= 0 && !(M3_ABL & 1)) gg.ph3();
            M3_PIN();
            if (CF::G1 && !(M3_ABL & 2)) M3Asm<DT>::acc(anxt1, w1, xf1[ks + 1]);
            M3_PIN();
            if (CF::GG && r >= 0 && !(M3_ABL & 1)) gg.template ph4<V8, EL>(acur1[r], acur1[r + 1], hn1, r);
            M3_PIN();
        };

        // ---- gemm2 of chunk i - 2 (4 steps) with the activation of panel 0 under it ----
        m3_read2<18>(fB, fa);
        m3_wait_lgkm<2>();
        step2(fA, 0, 0, 0);
        m3_read2<20>(fA, fa);
        m3_wait_lgkm<2>();
        step2(fB, 2, 2, 1);
        m3_read2<22>(fB, fa);
        m3_wait_lgkm<2>();
        step2(fA, 4, 4, 2);
        if (CF::G1) {
            m3_read2<0>(fA, fa);  // W1 k-steps 0, 1; then b1 of chunk i (C-layout register order): short-lived, read just ahead of its use
            m3_read<0>(bq[0], ta);
            m3_read<16>(bq[1], ta);
            m3_read<32>(bq[2], ta);
            m3_read<48>(bq[3], ta);
            m3_wait_lgkm<6>();
        } else {
            m3_wait_lgkm<0>();
            bq[0] = bq[1] = bq[2] = bq[3] = fA[0] = fA[1] = u32x4{0u, 0u, 0u, 0u};
        }
        step2(fB, 6, 6, 3);
        // ---- gemm1 of chunk i (8 steps; b1 is the C operand of the first MFMA of each panel) with the activation of panel 1 under it ----
        if (CF::G1) m3_read2<2>(fB, fa);
        m3_wait_lgkm<2>();
        {
            f32x16 bias;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                for (int e = 0; e < 4; ++e) bias[qd * 4 + e] = __uint_as_float(bq[qd][e]);
            const V8 w0 = __builtin_bit_cast(V8, fA[0]), w1 = __builtin_bit_cast(V8, fA[1]);
            if (CF::DMA && !M3_DMA_BARE) dma(nstage, nslot, 4);
            if (CF::G1 && !(M3_ABL & 2)) {
                M3Asm<DT>::first(anxt0, w0, xf0[0], bias);
                M3Asm<DT>::first(anxt1, w0, xf1[0], bias);
                // (a vector write to a register an in-flight MFMA still reads as its C operand is a software hazard -- 13 wait states for a 32x32
                //  MFMA --, and the compiler, which does not see an MFMA in the asm, is free to recycle the b1 registers right here)
                asm volatile("s_nop 7\n\ts_nop 6" ::: "memory");
            } else if (CF::G1) {
                anxt0 = bias;
                anxt1 = bias;
            }
            M3_PIN();
            if (CF::GG && !(M3_ABL & 1)) gg.ph1(acur1[8], acur1[9]);
            if (CF::GG && !(M3_ABL & 1)) gg.ph2();
            M3_PIN();
            if (CF::G1 && !(M3_ABL & 2)) M3Asm<DT>::acc(anxt0, w1, xf0[1]);
            M3_PIN();
            if (CF::GG && !(M3_ABL & 1)) gg.ph3();
            M3_PIN();
            if (CF::G1 && !(M3_ABL & 2)) M3Asm<DT>::acc(anxt1, w1, xf1[1]);
            M3_PIN();
            if (CF::GG && !(M3_ABL & 1)) gg.template ph4<V8, EL>(acur1[0], acur1[1], hn1, 0);
            M3_PIN();
        }
        if (CF::G1) m3_read2<4>(fA, fa);
        m3_wait_lgkm<2>();
        step1(fB, 2, -1, M3_DMA_BARE ? 0 : 5, M3_DMA_BARE ? 1 : -1);
        if (CF::G1) m3_read2<6>(fB, fa);
        m3_wait_lgkm<2>();
        step1(fA, 4, 2, -1, -1);
        if (CF::G1) m3_read2<8>(fA, fa);
        m3_wait_lgkm<2>();
        step1(fB, 6, -1, M3_DMA_BARE ? 2 : -1, M3_DMA_BARE ? 3 : -1);
        if (CF::G1) m3_read2<10>(fB, fa);
        m3_wait_lgkm<2>();
        step1(fA, 8, 4, -1, -1);
        if (CF::G1) m3_read2<12>(fA, fa);
        m3_wait_lgkm<2>();
        step1(fB, 10, -1, M3_DMA_BARE ? 4 : -1, M3_DMA_BARE ? 5 : -1);
        if (CF::G1) m3_read2<14>(fB, fa);
        m3_wait_lgkm<2>();
        step1(fA, 12, 6, -1, -1);
        m3_wait_lgkm<0>();
        step1(fB, 14, -1, -1, -1);
    };

    int slot = 0;
    constexpr int M3_NLOOP = M3_NCH;  // 64 uniform iterations (past stage 65 they re-request it: the wait count stays uniform)
#pragma unroll 1
    for (int i = 0; i < M3_NLOOP; i += 2) {
        iteration(M3It<24, true, true, true>{}, i, slot, a0, a1, b0, b1, h0a, h1a, h0b, h1b);
        slot = slot == M3_NS - 1 ? 0 : slot + 1;
        iteration(M3It<24, true, true, true>{}, i + 1, slot, b0, b1, a0, a1, h0b, h1b, h0a, h1a);
        slot = slot == M3_NS - 1 ? 0 : slot + 1;
    }
    static_assert(M3_NLOOP % 2 == 0 && M3_NIT == M3_NLOOP + 2 && M3_NS == 6, "the tail below is written out for these counts");
    // ---- the last two iterations, written out: no first GEMM (the x panels are dead: the residual rows are requested into their registers one iteration
    // ahead of the epilogue), no requests, and the last one no activation.  In flight at the top of 65: stage 65, 18 re-requests, then NEARLY of the 32 row requests (the rest at the epilogue's start: the allocator spilled a longer prefetch).
    auto next_slot = [&]() __attribute__((always_inline)) { slot = slot == M3_NS - 1 ? 0 : slot + 1; };
    __builtin_amdgcn_sched_barrier(0);
    // (buffer addressing: row (lane >> 5) + 2 v, 16-byte chunk lane & 31 = byte lane * 16 + v * 1024 of the wave's 32 KB of rows; the resource ends with
    //  the rows that exist, so a ragged last tile needs no clamps or predicates -- reads past it return 0, writes past it are dropped)
    // (the lane id is derived AGAIN here, opaquely: lane-derived values computed in the prologue and kept for the epilogue were spilled across the loop)
    int lane_e;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
    const int l31_e = lane_e & 31, half_e = lane_e >> 5;
    const int64_t nrow64 = p.M - mw0;
    const uint32_t nrow = (uint32_t)(nrow64 < 0 ? 0 : (nrow64 > 64 ? 64 : nrow64));
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.x) + mw0 * (M3_C * 2), 0, (int)(nrow * (M3_C * 2)), 0x00020000);
    const uint32_t roff = (uint32_t)(lane_e * 16);
    const int rrow0 = half_e;
    const uint32_t rcol = (uint32_t)(l31_e * 16);
    constexpr int NEARLY = DT == APAD_BF16 ? 16 : 0;  // row requests sent one iteration ahead (what the register allocator holds without spilling; f16's epilogue is tighter)
    u32x4 res[32];
    __builtin_amdgcn_sched_barrier(0);
    iteration(M3It<24, false, false, true>{}, M3_NLOOP, slot, a0, a1, b0, b1, h0a, h1a, h0b, h1b);  // GEGLU of chunk 63, gemm2 of 62
    next_slot();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int v = 0; v < NEARLY; ++v) res[v] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, roff, v * 1024, 0));
    __builtin_amdgcn_sched_barrier(0);
    iteration(M3It<18 + NEARLY, false, false, false>{}, M3_NLOOP + 1, slot, b0, b1, a0, a1, h0b, h1b, h0a, h1a);  // gemm2 of chunk 63
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: y + b2 -> storage type -> this wave's [64][256] tile in the dead ring -> + x (requested above) -> whole-row stores ----
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int v = NEARLY; v < 32; ++v) res[v] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, roff, v * 1024, 0));
    __syncthreads();
    uint8_t* const tile = smem + wave * (64 * M3_OROWB);
#pragma unroll
    for (int pn = 0; pn < 2; ++pn)
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& ya = pn == 0 ? y0[ct] : y1[ct];
                const float4 b4 = *reinterpret_cast<const float4*>(lb2 + ct * 32 + 8 * g + 4 * half_e);
                typename E::v4 yv;
                yv[0] = (typename E::elem)(ya[4 * g + 0] + b4.x);
                yv[1] = (typename E::elem)(ya[4 * g + 1] + b4.y);
                yv[2] = (typename E::elem)(ya[4 * g + 2] + b4.z);
                yv[3] = (typename E::elem)(ya[4 * g + 3] + b4.w);
                *reinterpret_cast<uint2*>(tile + (pn * 32 + l31_e) * M3_OROWB + (ct * 32 + 8 * g + 4 * half_e) * 2) = __builtin_bit_cast(uint2, yv);
            }
    // (a wave reads back only its own tile: its own LDS writes are ordered before its reads)
    {
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(p.out + mw0 * (M3_C * 2), 0, (int)(nrow * (M3_C * 2)), 0x00020000);
#pragma unroll
        for (int v = 0; v < 32; ++v) {
            float f[8], r[8];
            unpack8<DT>(*reinterpret_cast<const uint4*>(tile + (rrow0 + 2 * v) * M3_OROWB + rcol), f);
            unpack8<DT>(__builtin_bit_cast(uint4, res[v]), r);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += r[e];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pack8<DT>(f)), rout, roff, v * 1024, 0);
        }
    }
}

// ---- apad_mlp_pack: W1 [8C][C], b1 [8C], W2 [C][4C] -> the stream of M3_NIT stages + the fp32 bias table ----
template <int DT>
__global__ void mlp3_pack_kernel(const uint8_t* w1, const uint8_t* b1, const uint8_t* w2, uint8_t* wpk, float* b1p) {
    using elem = typename ET<DT>::elem;
    const int64_t total = (int64_t)M3_NIT * (M3_STAGE / 2);  // elements
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / (M3_STAGE / 2)), r = (int)(e % (M3_STAGE / 2));
        const int frag = r / 512, rem = r % 512, lane = rem / 8, j = rem % 8, hf = lane >> 5, l31 = lane & 31;
        elem v = (elem)0.f;
        if (frag < 16) {  // W1 of chunk i: rows 0..15 the value units, 16..31 the gate units; k = frag * 16 + half * 8 + j
            if (i < M3_NCH) {
                const int row = l31 < 16 ? i * 16 + l31 : M3_HID + i * 16 + (l31 - 16);
                v = reinterpret_cast<const elem*>(w1)[(int64_t)row * M3_C + frag * 16 + hf * 8 + j];
                if (l31 < 16) v = (elem)((float)v * m3_value_scale<DT>());  // (the value rows carry GEGLU's 1/2: M3GegluT<PRE>)
            }
        } else {  // W2 of chunk i - 2, output-column tile frag - 16: the k slot (half, j) holds hidden unit (j & 3) + 8 (j >> 2) + 4 half
            const int c = i - 2, ct = frag - 16;
            if (c >= 0 && c < M3_NCH) v = reinterpret_cast<const elem*>(w2)[(int64_t)(ct * 32 + l31) * M3_HID + c * 16 + (j & 3) + 8 * (j >> 2) + 4 * hf];
        }
        reinterpret_cast<elem*>(wpk)[e] = v;
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < M3_NIT * 32; t += gridDim.x * blockDim.x) {
        const int i = t / 32, hf = (t % 32) / 16, r = t % 16, u = ((r & 7) & 3) + 8 * ((r & 7) >> 2) + 4 * hf;
        float v = 0.f;
        if (i < M3_NCH && b1 != nullptr) v = (float)reinterpret_cast<const elem*>(b1)[(r < 8 ? 0 : M3_HID) + i * 16 + u] * (r < 8 ? m3_value_scale<DT>() : 1.0f);
        b1p[t] = v;
    }
}

inline bool m3_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int DT, bool LN> int mlp3_launch(const Mlp3P& p, hipStream_t s) {
    auto kern = mlp3_kernel<DT, LN>;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), M3_LDS, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)((p.M + 255) / 256)), dim3(256), M3_LDS, s, p);
    return apad_check_launch("apad_geglu_mlp_packed");
}

}  // namespace

extern "C" int64_t apad_mlp_packed_bytes(int32_t C) { return C == M3_C ? (int64_t)M3_NIT * M3_STAGE : -1; }
extern "C" int64_t apad_mlp_packed_bias_floats(int32_t C) { return C == M3_C ? (int64_t)M3_NIT * 32 : -1; }

extern "C" int apad_mlp_pack(const void* w1, const void* b1, const void* w2, void* w_packed, float* b1_packed, int32_t C, int32_t dtype, void* stream) {
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_mlp_pack: dtype %d not supported", dtype);
    APAD_CHECK(w1 && w2 && w_packed && b1_packed, "apad_mlp_pack: null operand");
    if (C != M3_C) {
        apad_set_error("apad_mlp_pack: C=%d outside the kernel envelope (256)", C);
        return -3;
    }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == APAD_BF16)
        hipLaunchKernelGGL(mlp3_pack_kernel<APAD_BF16>, dim3(512), dim3(256), 0, s, (const uint8_t*)w1, (const uint8_t*)b1, (const uint8_t*)w2, (uint8_t*)w_packed, b1_packed);
    else
        hipLaunchKernelGGL(mlp3_pack_kernel<APAD_F16>, dim3(512), dim3(256), 0, s, (const uint8_t*)w1, (const uint8_t*)b1, (const uint8_t*)w2, (uint8_t*)w_packed, b1_packed);
    return apad_check_launch("apad_mlp_pack");
}

extern "C" int apad_geglu_mlp_packed(const apad_mlp_desc* d, const void* w_packed, const float* b1_packed, void* stream) {
    APAD_CHECK(d != nullptr, "apad_geglu_mlp_packed: null descriptor");
    APAD_CHECK(d->dtype == APAD_BF16 || d->dtype == APAD_F16, "apad_geglu_mlp_packed: dtype %d not supported", d->dtype);
    APAD_CHECK(d->x && w_packed && b1_packed && d->out && d->M > 0, "apad_geglu_mlp_packed: null operand / empty problem");
    APAD_CHECK(m3_al16(d->x) && m3_al16(w_packed) && m3_al16(b1_packed) && m3_al16(d->out) && m3_al16(d->ln_gamma) && m3_al16(d->ln_beta),
               "apad_geglu_mlp_packed: pointers must be 16-byte aligned");
    if (d->C != M3_C) {
        apad_set_error("apad_geglu_mlp_packed: C=%d outside the kernel envelope (256)", d->C);
        return -3;
    }
    const bool ln = d->ln_gamma != nullptr;
    if (ln) APAD_CHECK(d->ln_beta != nullptr, "apad_geglu_mlp_packed: LayerNorm needs gamma and beta");
    Mlp3P p;
    p.x = (const uint8_t*)d->x; p.gamma = (const uint8_t*)d->ln_gamma; p.beta = (const uint8_t*)d->ln_beta;
    p.wpk = (const uint8_t*)w_packed; p.b1p = b1_packed; p.b2 = (const uint8_t*)d->b2;
    p.out = (uint8_t*)d->out; p.M = d->M; p.eps = d->ln_eps;
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == APAD_BF16) return ln ? mlp3_launch<APAD_BF16, true>(p, s) : mlp3_launch<APAD_BF16, false>(p, s);
    return ln ? mlp3_launch<APAD_F16, true>(p, s) : mlp3_launch<APAD_F16, false>(p, s);
}
