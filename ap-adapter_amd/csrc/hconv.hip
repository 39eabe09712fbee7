// Halo-resident implicit-GEMM 3x3 convolution (stride 1, zero padding 1; optionally over a nearest-up-sampled source) for the resnet /
// up-sampler convolutions of the UNet levels whose image width is a power of two <= 16 (modeling_audioldm2.py: ResnetBlock2D conv1 /
// conv2, Upsample2D.conv) -- apad_gemm selects it for EVERY row count when the caller supplies the packed weight form
// (apad_gemm_desc::w_halo, apad_conv_halo_pack), so a pixel's result does not depend on the batch it rides in.
//
// What the im2col form (cgemm.hip) spends its time on, measured with its ablation builds (profiles/r06_cgemm_ablation.txt): the LDS-DMA
// stream alone takes 66 of the kernel's 70 us at 256 -> 256 channels / 1000 pixels -- every one of the nine filter taps re-fetches its
// A tile from L2, and between two passes over the same activation lines an XCD's workgroups stream 8 MB through its 4 MB L2, so the nine
// passes are served by the Infinity Cache, not by L2.  Here the A operand is fetched ONCE per 64-channel chunk:
//   * a workgroup owns BM consecutive output pixels (= BM / W whole image rows) and keeps, per 64-channel chunk, their (rows + 2) x W
//     input pixels in LDS: the nine taps read the same halo tile at nine uniform pixel shifts.  Zero padding: rows above / below a sample
//     are zero rows of the tile (one separator row between two samples of a tile, written as zeros by the DMA's range check); the left /
//     right image border re-aims the border lanes of the dx = -1 / +1 taps at a 256-byte zero region (the zero slot of the lane's own bank)
//   * summation order: 64-channel chunk (outer), tap, four 16-deep MFMA steps -- this kernel's own order, identical for every tile shape
//   * the halo tile is double buffered (the next chunk's five 1 KB DMA pieces per wave are issued over the first stages of the current
//     chunk); weights stream through a four-stage LDS ring from the packed form [chunk][tap][32-channel half][N][32] (contiguous,
//     pre-swizzled: one DMA piece = 1 KB of memory), three stages ahead, counted vmcnt, ONE raw barrier per stage
//   * bank-conflict-free fragment reads on both operands: pixel records of 128 bytes with the 16-byte slot XOR-ed by (pixel >> 1) & 7,
//     weight records of 64 bytes XOR-ed by (n >> 2) & 3 (any 16 consecutive pixels / rows, shifted by any tap, cover all 16 slot banks)
//   * fragment reads are software-pipelined in registers (the reads of MFMA step k + 1 are issued in front of the MFMAs of step k, also
//     across the stage barrier: the barrier of stage t certifies stage t + 1)
//   * MFMA roles swapped against cgemm.hip (weights are the A operand): a lane's accumulators are 16 output channels of ONE pixel, so
//     the epilogue packs, exchanges half-wave pairs (v_permlane32_swap) and stores 16 bytes per lane straight from registers -- no LDS
//     round trip, no barrier
#include <stdlib.h>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

#ifndef HC_ABL
#define HC_ABL 0  // ablation bits for timing-only probe builds (tools/ab_build.sh): 1 no DMA in the loop, 2 no fragment reads, 4 no MFMAs, 8 no stage barrier
#endif
constexpr uint32_t H_OOB = 0x80000000u;

struct HcP {
    const uint8_t* a;
    const uint8_t* wp;
    uint8_t* out;
    const uint8_t* bias;
    const uint8_t* residual;
    const uint8_t* rg;
    const int32_t* step_ptr;
    int64_t ldo, ldr, ld_rg, rows_per_group;
    int32_t M, N, Cin;
    int32_t H, W, Wlog;  // the output image (= the source image unless up-sampled)
    int32_t Hs, Ws;      // the source image
    int32_t Btot;
    int32_t m_tiles, n_tiles, nchunks;
    int32_t ksplit;   // K slices per output tile (whole 64-channel chunks each); > 1: fp32 partial slabs, summed in slice order by hconv_reduce_kernel
    float* partial;   // [ksplit][M][N]
    uint32_t a_bytes, w_bytes;
    unsigned long long* trace;  // (probe builds, HC_TRACE: 32 s_memtime stamps per workgroup; tools/hconv_trace.py)
};
#ifndef HC_TRACE
#define HC_TRACE 0
#endif
#define HC_STAMP(k)                                                                   \
    do {                                                                              \
        if (HC_TRACE && p.trace && tid == 0 && (k) < 32) p.trace[blockIdx.x * 32 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t h_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

#define H_FENCE() asm volatile("" ::: "memory")
#define H_BARRIER()                          \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        H_FENCE();                           \
        __builtin_amdgcn_s_barrier();        \
        H_FENCE();                           \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

template <int N_> __device__ __forceinline__ void h_wait_vm() {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N_ == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N_ == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else static_assert(N_ == 0, "add the count");
}

template <int... S, class F> __device__ __forceinline__ void h_for_each(std::integer_sequence<int, S...>, F&& f) {
    (f(std::integral_constant<int, S>{}), ...);
}

#ifndef HC_AGPR
#define HC_AGPR 0  // 1: accumulators in the AGPR half of the register file (inline-asm MFMAs); 0: the compiler's VGPR-form MFMAs
#endif
// acc += a . b with the accumulator block in AGPRs: the 16-register C read / D write-back of every MFMA then goes through the accumulator file's
// ports and leaves the VGPR ports to the fragment reads coming back from LDS.  (a, b: fragment registers written by ds_read_b128, waited for with
// lgkmcnt by the caller -- no VALU-write -> MFMA-read hazard; back-to-back MFMAs on the same accumulator need no wait states)
template <int DT> __device__ __forceinline__ void h_mfma_acc(f32x16& acc, const unsigned int __attribute__((ext_vector_type(4)))& a,
                                                              const unsigned int __attribute__((ext_vector_type(4)))& b) {
    if constexpr (DT == APAD_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int OFF> __device__ __forceinline__ void h_read(u32x4& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}

// WAVES_M x WAVES_N waves of (MI x 32 pixels) x (NJ x 32 channels); KS = MFMA steps (16 channels each) per weight stage (2: one 32-channel
// half of a tap, 4: a whole tap of the 64-channel chunk)
template <int WAVES_M_, int WAVES_N_, int MI_, int NJ_, int KS_> struct HcT {
    static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, MI = MI_, NJ = NJ_, KS = KS_;
    static constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    static constexpr int BM = WAVES_M * MI * 32, BN = WAVES_N * NJ * 32;
    static constexpr int NLD = 5;                      // halo DMA pieces per wave and chunk
    static constexpr int NPX = NLD * NW * 8;           // pixel records of a halo buffer (320 / 160)
    static constexpr int ABUF = NPX * 128;
    static constexpr int SUBB = BN * 64;               // one 32-channel half of a tap: BN rows x 64 bytes
    static constexpr int STAGE_B = SUBB * (KS / 2);
    static constexpr int NSTG = 4;
    static constexpr int PBW = STAGE_B / 1024 / NW;    // weight DMA pieces per wave and stage
    static constexpr int SPT = 4 / KS;                 // stages per tap
    static constexpr int SPC = 9 * SPT;                // stages per 64-channel chunk
    // LDS: [halo buffer 0][ring slots 0 .. 3][halo buffer 1][zero region]: ring slot 3 and halo buffer 1 -- both idle while a tile's epilogue
    // runs -- are adjacent, the epilogue's transposition tiles live there
    static constexpr int OFF_B = ABUF, OFF_A1 = OFF_B + NSTG * STAGE_B, OFF_Z = OFF_A1 + ABUF, SMEM = OFF_Z + 256;
    static constexpr int OFF_STG = OFF_B + (NSTG - 1) * STAGE_B, STG_BYTES = STAGE_B + ABUF;
    static_assert(STAGE_B % (1024 * NW) == 0 && (KS == 2 || KS == 4) && SMEM <= 160 * 1024, "tile shape");
    static_assert(NJ * 2048 + SUBB < 65536, "immediate offsets of the weight fragment reads");
};

template <int DT, class T>
__global__ __launch_bounds__(T::NT) void hconv_kernel(HcP p) {
    constexpr int MI = T::MI, NJ = T::NJ, KS = T::KS, NW = T::NW, BM = T::BM, BN = T::BN, NLD = T::NLD, PBW = T::PBW, SPT = T::SPT, SPC = T::SPC;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using E = ET<DT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
    const int half = lane >> 5, l31 = lane & 31;

    const int W = p.W, H = p.H, Wlog = p.Wlog;
    const int R = BM >> Wlog;  // image rows of a tile
    const __amdgpu_buffer_rsrc_t ra = h_rsrc(p.a, p.a_bytes), rw = h_rsrc(p.wp, p.w_bytes);
    const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr)smem;

    // ---- per tile: origin, halo DMA sources, the lanes' pixel records.  Persistent workgroups: tile tl, tl + gridDim.x, ... ----
    int m0 = 0, n0 = 0, c_begin = 0, c_end = 0, kslice = 0;
    uint32_t aoff[5];   // piece l of this wave fills pixel records (l NW + wave) 8 .. + 8; lane -> (record, 16-byte slot); the slot holds source
                        // chunk slot ^ ((record >> 1) & 7).  Record pp <-> (tile row j = pp / W, x): virtual row v = y0 - 1 + j in a coordinate
                        // where every sample owns H rows + one separator
    uint32_t pbase[4];  // this lane's output pixel of MFMA tile i sits at record pbase[i]; tap (dy, dx) reads record pbase + dy W + dx
                        // (MI used.  Fixed bounds: a template-dependent array bound captured by the lambdas below loses the kernel's host stub -- hipcc 7.2)
    auto setup_tile = [&](int tl_ks) {
        // the K slice (fastest index) and its range of 64-channel chunks
        const int tl = tl_ks / p.ksplit;
        kslice = tl_ks - tl * p.ksplit;
        c_begin = kslice * p.nchunks / p.ksplit;
        c_end = (kslice + 1) * p.nchunks / p.ksplit;
        // XCD-aware tile order (speed only): all N-tiles of one M-tile share tl % 8 -- one XCD's L2 serves their common halo
        int mt, nt;
        {
            const int nN = p.n_tiles, nM = p.m_tiles;
            const int full = (nM / 8) * 8 * nN;
            if (tl < full) {
                const int g = tl / (8 * nN), rem = tl - g * 8 * nN;
                nt = rem >> 3;
                mt = g * 8 + (rem & 7);
            } else {
                const int rem = tl - full, tail = nM - (nM / 8) * 8;
                nt = rem / tail;
                mt = (nM / 8) * 8 + rem - nt * tail;
            }
        }
        m0 = mt * BM;
        n0 = nt * BN;
        const int g0 = mt * R;  // the tile's first global row (sample * H + y)
        const int b0 = g0 / H, y0 = g0 - b0 * H;
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            const int pp = (l * NW + wave) * 8 + (lane >> 3);
            const int slot = lane & 7;
            const int j = pp >> Wlog, x = pp & (W - 1);
            const int v = y0 - 1 + j;
            const int q = v >= 0 ? v / (H + 1) : 0, r = v - q * (H + 1);
            const bool valid = v >= 0 && r != H && b0 + q < p.Btot;
            const int sy = p.Hs == H ? r : (r * p.Hs) / H, sx = p.Ws == W ? x : (x * p.Ws) / W;
            const uint32_t src = (uint32_t)(((b0 + q) * p.Hs + sy) * p.Ws + sx) * (uint32_t)(p.Cin * 2) + (uint32_t)((slot ^ ((pp >> 1) & 7)) << 4);
            aoff[l] = valid ? src : H_OOB;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = (wm * MI + i) * 32 + l31;
            const int ir = ml >> Wlog, x = ml & (W - 1);
            const int nsep = (y0 + ir) / H;
            pbase[i] = (uint32_t)(((1 + ir + nsep) << Wlog) + x);
        }
    };
    auto issue_a = [&](int l, int buf, int chunk) {  // (wave-uniform arguments)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(smem + buf * T::OFF_A1 + (l * NW + wave) * 1024), 16, aoff[l], chunk * 128, 0, 0);
    };
    // weight stage (chunk c, stage s of the chunk) -> ring slot: its KS / 2 sub-blocks [N][64 bytes] of the packed form
    const uint32_t lane16 = (uint32_t)(lane * 16);
    auto issue_b = [&](int c, int s, int slot, int i) {  // piece i (of PBW) of this wave
        const int sub0 = (c * 9 + s / SPT) * 2 + (s % SPT) * (KS / 2);  // first 32-channel sub-block of the stage
        const int q = wave * PBW + i;                                   // piece of the stage
        const int sb = q / (T::SUBB / 1024), qq = q - sb * (T::SUBB / 1024);
        const int soff = ((sub0 + sb) * p.N + n0) * 64 + qq * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + T::OFF_B + slot * T::STAGE_B + q * 1024), 16, lane16, soff, 0, 0);
    };

    // ---- fragment addressing ----
    const bool x_first = (l31 & (W - 1)) == 0, x_last = (l31 & (W - 1)) == W - 1;
    // the zero region: 256 bytes = all 16 slot banks.  A border lane reads the zero slot in ITS OWN bank (address bits 4..7 kept): re-aimed at one fixed
    // slot it collided with the lane that owns that bank -- 20 % of the kernel's LDS cycles were bank-conflict cycles (profiles/r06_pmc_hconv256_v6.txt)
    const uint32_t zbase = lds0 + (uint32_t)T::OFF_Z;
    // weights: row n of the tile, k-chunk (ks' 2 + half) of a 64-byte record, slot XOR-ed by (n >> 2) & 3
    uint32_t bfo[2];
    {
        const int nl = wn * NJ * 32 + l31;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) bfo[k2] = lds0 + (uint32_t)T::OFF_B + (uint32_t)(nl * 64 + (((k2 * 2 + half) ^ ((nl >> 2) & 3)) << 4));
    }

    f32x16 acc[4][4];
    u32x4 fa[2][4] = {}, fb[2][4] = {};
    // per MFMA tile: LDS address of (this lane's pixel shifted by a tap, k-chunk `half`) in a halo buffer; a0: the current tap, a0n: the next one
    uint32_t a0[4], a0n[4];
    auto tap_address = [&](int i, int tap, int buf) -> uint32_t {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const uint32_t ps = pbase[i] + (uint32_t)(dy * W + dx);
        uint32_t a = lds0 + (uint32_t)(buf * T::OFF_A1) + (ps << 7) + ((((ps >> 1) & 7) ^ (uint32_t)half) << 4);
        if (dx < 0) a = x_first ? (zbase | (a & 0xF0u)) : a;
        if (dx > 0) a = x_last ? (zbase | (a & 0xF0u)) : a;
        return a;
    };
    // ONE fragment read of MFMA step KSI (0..3 of the chunk; k-chunk 2 KSI + half) into register set SET: IDX < MI the pixel tile IDX (from
    // a0, or a0n when NEXT), else the weight tile IDX - MI at bb (= bfo[KSI & 1] + ring slot)
    bool abl_reads_off = false;  // (HC_ABL & 2: the prologue still fills both register sets with real data, so the MFMAs draw their real power)
    auto read_one = [&](auto set_tag, auto ks_tag, auto idx_tag, auto next_tag, uint32_t bb) {
        constexpr int SET = decltype(set_tag)::value, KSI = decltype(ks_tag)::value, IDX = decltype(idx_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value != 0;
        if ((HC_ABL & 2) && abl_reads_off) return;
        if constexpr (IDX < MI) {
            h_read<0>(fa[SET][IDX], (NEXT ? a0n[IDX] : a0[IDX]) ^ (uint32_t)(KSI << 5));
        } else if constexpr (IDX < MI + NJ) {
            constexpr int J = IDX - MI, SB = (KSI % KS) >> 1;  // sub-block of the stage
            h_read<SB * T::SUBB + J * 2048>(fb[SET][J], bb);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    constexpr int NMF = MI * NJ, NRD = MI + NJ;
    // placement inside an MFMA step: the NRD fragment reads of the next step go two at a time behind MFMAs 0, 1, ..; the DMA requests of a stage
    // (halo piece first, then the PBW weight pieces) behind the MFMAs after them.  An LDS-DMA instruction holds its wave for ~100 cycles
    // (tools/ubench/dmaissue.hip): the two waves of a SIMD (w, w + 4) issue theirs in different MFMA steps, so the partner's MFMAs fill the pipe
    constexpr int RSLOTS = (NRD + 1) / 2;
    constexpr int DSLOT0 = RSLOTS < NMF - 1 ? RSLOTS : NMF - 2;
    const int grp = wave >> 2;

    // ---- a tile's prologue requests: halo chunk 0 -> buffer 0, weight stages 0 .. 2 -> ring slots 0 .. 2 ----
    auto issue_prologue = [&]() {
#pragma unroll
        for (int l = 0; l < NLD; ++l) issue_a(l, 0, c_begin);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < PBW; ++i) issue_b(c_begin, s, s, i);
    };
    if (tid < 16) *reinterpret_cast<uint4*>(smem + T::OFF_Z + tid * 16) = make_uint4(0, 0, 0, 0);  // the zero region
    const int ntiles = p.m_tiles * p.n_tiles * p.ksplit;
    HC_STAMP(0);
    setup_tile(blockIdx.x);
    issue_prologue();
    HC_STAMP(1);
    int tstamp = 2;
#pragma unroll 1
    for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int cm0 = m0, cn0 = n0, cks = kslice, cb = c_begin, ce = c_end;  // (the next tile's setup overwrites them before this tile's epilogue)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
    h_wait_vm<2 * PBW>();  // the halo and stage 0 have landed (this wave's pieces; whatever the previous tile's epilogue left in flight is older)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    H_BARRIER();
#pragma unroll
    for (int i = 0; i < MI; ++i) a0[i] = a0n[i] = tap_address(i, 0, 0);
    h_for_each(std::make_integer_sequence<int, NRD>{}, [&](auto idx_tag) { read_one(I0{}, I0{}, idx_tag, I0{}, bfo[0]); });
    HC_STAMP(tstamp);  // prologue waited for, first reads requested
    if (HC_ABL & 2) {
        h_for_each(std::make_integer_sequence<int, NRD>{}, [&](auto idx_tag) { read_one(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, idx_tag, I0{}, bfo[1]); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        abl_reads_off = true;
    }

    // ---- main loop: chunks (run time) x SPC stages (unrolled) ----
    int slot = 0;  // ring slot of the current stage
#pragma unroll 1
    for (int c = cb; c < ce; ++c) {
        const bool last = c + 1 == ce;
        const int buf = (c - cb) & 1;
        auto stage = [&](auto s_tag) {
            constexpr int S = decltype(s_tag)::value;
            constexpr int TAP = S / SPT, KS0 = (S % SPT) * KS;  // first MFMA step (of the chunk's four) of this stage
            // (1) stage t + 1 has landed: at most what the previous iteration issued may be outstanding (the halo piece goes first in an
            //     iteration, so a halo piece is covered two iterations after its issue)
            if (HC_TRACE && tl == (int)blockIdx.x && c == cb + 1 && S < 12) HC_STAMP(19 + S);  // (probe: the stages of the tile's SECOND chunk)
            constexpr bool prevA = S >= 1 && S - 1 < NLD;  // the previous iteration issued a halo piece (not in the last chunk)
            if (!last) h_wait_vm<PBW + (prevA ? 1 : 0)>();
            else if (S == 0 || S - 1 + 3 < SPC) h_wait_vm<PBW>();
            else h_wait_vm<0>();
            if (!(HC_ABL & 128)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(HC_ABL & 8)) H_BARRIER();
            // (2) the MFMA steps.  Behind single MFMAs: the fragment reads of step k + 1 (of the next stage behind the last step: the barrier
            //     above certified it), the next tap's addresses, and this iteration's requests -- halo piece S of the next chunk, weight stage
            //     t + 3 (into the slot read in iteration t - 1)
            const int nslot = (slot + 1) & 3;
            const bool dma_a = S < NLD && !last, dma_b = S + 3 < SPC || !last;
            h_for_each(std::make_integer_sequence<int, KS>{}, [&](auto kk_tag) {
                constexpr int KK = decltype(kk_tag)::value, SET = KK & 1;
                constexpr bool SAME = KK + 1 < KS;                  // the next MFMA step lies in this stage
                constexpr int NS = (S + 1) % SPC;                   // the next stage (of this or the next chunk)
                constexpr bool NEWTAP = !SAME && NS % SPT == 0;     // the prefetch crosses into the next tap
                constexpr int PKSI = SAME ? KS0 + KK + 1 : (NS % SPT) * KS;
                constexpr bool TAPCALC = S % SPT == SPT - 1 && KK == 0;  // the next tap's addresses: first step of a tap's last stage
                constexpr int NTAP = (TAP + 1) % 9;
                const bool pf = SAME || S + 1 < SPC || !last;
                const uint32_t bb = bfo[PKSI & 1] + (uint32_t)((SAME ? slot : nslot) * T::STAGE_B);
                const bool dma_here = !(HC_ABL & 1) && grp == (KK == 0 ? 0 : (KK == KS / 2 ? 1 : 2));
                h_for_each(std::make_integer_sequence<int, NMF>{}, [&](auto idx_tag) {
                    constexpr int IDX = decltype(idx_tag)::value;
                    if (!(HC_ABL & 4)) {
                        if constexpr (HC_AGPR) h_mfma_acc<DT>(acc[IDX / MI][IDX % MI], fb[SET][IDX / MI], fa[SET][IDX % MI]);
                        else acc[IDX / MI][IDX % MI] = E::mfma32(__builtin_bit_cast(typename E::v8, fb[SET][IDX / MI]),
                                                                 __builtin_bit_cast(typename E::v8, fa[SET][IDX % MI]), acc[IDX / MI][IDX % MI]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (IDX < RSLOTS) if (pf) {
                        read_one(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, PKSI>{}, std::integral_constant<int, 2 * IDX>{},
                                 std::integral_constant<int, NEWTAP>{}, bb);
                        read_one(std::integral_constant<int, SET ^ 1>{}, std::integral_constant<int, PKSI>{}, std::integral_constant<int, 2 * IDX + 1>{},
                                 std::integral_constant<int, NEWTAP>{}, bb);
                    }
                    if constexpr (TAPCALC && IDX < MI) a0n[IDX] = tap_address(IDX, NTAP, TAP + 1 < 9 ? buf : buf ^ 1);
                    if (dma_here) {
                        if (IDX == DSLOT0 && dma_a) issue_a(S, buf ^ 1, c + 1);
                        if (IDX >= DSLOT0 && IDX - DSLOT0 < PBW && dma_b) {
                            if (S + 3 < SPC) issue_b(c, S + 3, (slot + 3) & 3, IDX - DSLOT0);
                            else issue_b(c + 1, S + 3 - SPC, (slot + 3) & 3, IDX - DSLOT0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (SAME && !(HC_ABL & 64)) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (NEWTAP) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) a0[i] = a0n[i];
                }
            });
            slot = nslot;
        };
        h_for_each(std::make_integer_sequence<int, SPC>{}, stage);
    }

    // every wave is past its last fragment read (the reads of a step are waited for inside the step) and its last DMA wait: behind this barrier
    // the halo buffers and the ring are free.  The NEXT tile's prologue requests go out first; this tile's epilogue runs under their latency, and
    // its stores drain under the next tile's main loop
    // (the epilogue's bias / table-row loads are requested here: their latency runs under the barrier and the next tile's setup)
    const int64_t step = p.step_ptr ? (int64_t)*p.step_ptr : 0;
    using elem = typename E::elem;
    const bool rg_table = p.rg && p.rows_per_group >= p.M;  // one time-embedding row for every pixel (the denoise loop's table form)
    const bool rg_rows = p.rg && !rg_table;                  // the per-sample form: row m / rows_per_group
    constexpr int EROW = NJ * 64 + 16;          // staged row: NJ x 32 channels + 16 bytes of padding (bank spread of the 8-byte writes)
    constexpr int LPR = NJ * 4;                 // lanes per staged row on the way out (16 bytes each)
    constexpr int RPI = 64 / LPR;               // rows per store instruction
    static_assert(32 * EROW * NW <= T::STG_BYTES, "the epilogue tiles of all waves fit ring slot 3 + halo buffer 1");
    const uint32_t stg = lds0 + (uint32_t)(T::OFF_STG + wave * (32 * EROW));
    const uint32_t stg_w = stg + (uint32_t)(l31 * EROW + 8 * half), stg_r = stg + (uint32_t)((lane / LPR) * EROW + (lane % LPR) * 16);
    const int cn_w = cn0 + wn * NJ * 32;        // first column of this wave
    uint2 braw[4][4], traw[4][4];               // [j][g]: bias / table row of this lane's 4 channels of a group, in the storage type (two registers a group)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = cn_w + j * 32 + 8 * g + 4 * half;
            braw[j][g] = traw[j][g] = make_uint2(0, 0);
            if (p.bias && p.ksplit == 1) braw[j][g] = *reinterpret_cast<const uint2*>(p.bias + (int64_t)n * 2);
            if (rg_table && p.ksplit == 1) traw[j][g] = *reinterpret_cast<const uint2*>(p.rg + (step * p.ld_rg + n) * 2);
        }
    abl_reads_off = false;
    HC_STAMP(tstamp + 1);  // main loop done
    H_BARRIER();
    HC_STAMP(tstamp + 2);
    if constexpr (HC_AGPR) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs' results before the compiler's v_accvgpr_read
    // the next tile's setup + requests go out first: their DMA latency runs under this tile's epilogue.  (The 256 x 256 form keeps 164 bytes per lane of
    // loop-invariant addressing state in scratch -- written in the prologue, reloaded here: profiles/r06_pmc_hconv256_v6.txt shows it as 22 MB of WRITE_SIZE
    // beside the 32.8 MB output; moving this setup behind the epilogue or packing the results first did not remove it -- the main loop's 128 accumulators +
    // 48 fragment registers + addressing are what fill the file.)
    if (tl + (int)gridDim.x < ntiles) {
        setup_tile(tl + (int)gridDim.x);
        issue_prologue();
    }
    HC_STAMP(tstamp + 3);  // next tile's setup + requests
    // ---- epilogue: (acc + bias) + time-embedding row -> storage type -> transposed through a wave-private LDS tile (ring slot 3 + halo buffer 1: idle until
    //      the next tile's first stage barrier) -> (+ residual) -> full-line stores.  Lane (pixel l31, half) holds channels 8 g + 4 half + (0..3) of a
    //      32-channel block in accumulators 4 g .. 4 g + 3: one ds_write_b64 per group; a row of the wave's NJ x 32 channels is then read back by NJ x 4
    //      lanes as 16-byte pieces, so one store instruction writes whole rows of the wave's column range ----
    if (p.ksplit > 1) {
        // a K slice: the raw fp32 accumulators go to this slice's slab (lane: 4 consecutive channels per group = one 16-byte store); bias, time
        // embedding, rounding and the residual belong to hconv_reduce_kernel, which sums the slabs in slice order
        float* slab = p.partial + (int64_t)cks * p.M * p.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = cm0 + (wm * MI + i) * 32 + l31;
            if (m < p.M) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                        *reinterpret_cast<f32x4*>(slab + (int64_t)m * p.N + cn_w + j * 32 + 8 * g + 4 * half) = v;
                    }
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int mrow0 = cm0 + (wm * MI + i) * 32;  // first pixel of the block
        // the residual rows of the block are requested first (row layout: lane -> (row lane / LPR + RPI k, piece lane % LPR))
        uint4 rres[32 / RPI] = {};
        if (p.residual) {
#pragma unroll
            for (int k = 0; k < 32 / RPI; ++k) {
                int m = mrow0 + lane / LPR + RPI * k;
                m = m < p.M ? m : p.M - 1;
                rres[k] = *reinterpret_cast<const uint4*>(p.residual + ((int64_t)m * p.ldr + cn_w + (lane % LPR) * 8) * 2);
            }
        }
        if (HC_TRACE && ntiles <= (int)gridDim.x) HC_STAMP(8 + 2 * i);
        const int mp = mrow0 + l31 < p.M ? mrow0 + l31 : p.M - 1;
        const int64_t grp = rg_rows ? mp / p.rows_per_group + step : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 tr = traw[j][g];
                if (rg_rows) tr = *reinterpret_cast<const uint2*>(p.rg + (grp * p.ld_rg + cn_w + j * 32 + 8 * g + 4 * half) * 2);
                const typename E::v4 b4 = __builtin_bit_cast(typename E::v4, braw[j][g]), t4 = __builtin_bit_cast(typename E::v4, tr);
                typename E::v4 h4;
#pragma unroll
                for (int e = 0; e < 4; ++e) h4[e] = (elem)((acc[j][i][4 * g + e] + (float)b4[e]) + (float)t4[e]);
                // (inline asm: an LDS access the compiler can see is ordered behind the next tile's DMA in flight -- s_waitcnt vmcnt(0))
                const uint2 u2 = __builtin_bit_cast(uint2, h4);
                asm volatile("ds_write_b64 %0, %1" ::"v"(stg_w + (uint32_t)((j * 32 + 8 * g) * 2)), "v"(u2) : "memory");
            }
        // wave-private tile: no barrier, and no wait between this wave's writes and its reads (the LDS executes one wave's instructions in issue order)
        u32x4 orow[32 / RPI];
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(orow[k]) : "v"(stg_r + (uint32_t)(RPI * k * EROW)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 32 / RPI; ++k) {
            const int rl = lane / LPR + RPI * k, m = mrow0 + rl;
            uint4 o = __builtin_bit_cast(uint4, orow[k]);
            if (p.residual) {
                float f[8], rr[8];
                unpack8<DT>(o, f);
                unpack8<DT>(rres[k], rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += rr[e];
                o = pack8<DT>(f);
            }
            if (m < p.M && !((HC_ABL & 32) && p.Btot > 0))  // (ablation 32: the stores stay in the code, no lane executes them)
                *reinterpret_cast<uint4*>(p.out + ((int64_t)m * p.ldo + cn_w + (lane % LPR) * 8) * 2) = o;
        }
        if (HC_TRACE && ntiles <= (int)gridDim.x) HC_STAMP(8 + 2 * i + 1);
    }
    }
    HC_STAMP(tstamp + 4);  // epilogue issued
    tstamp += 5;
    }  // tiles
}

// [Cout][ky][kx][Cin] -> [Cin / 64][tap][2][Cout][32 channels], the 16-byte slot s of a 64-byte record holding channels ((s ^ ((n >> 2) & 3)) 8 ..
template <typename TE>
__global__ void hconv_pack_kernel(const TE* __restrict__ w, TE* __restrict__ out, int N, int Cin, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte slot
    if (idx >= total) return;
    const int s = (int)(idx & 3);
    int64_t r = idx >> 2;
    const int n = (int)(r % N);
    r /= N;
    const int h = (int)(r & 1);
    r >>= 1;
    const int tap = (int)(r % 9), c = (int)(r / 9);
    const int ch = c * 64 + h * 32 + ((s ^ ((n >> 2) & 3)) << 3);
    const uint4 v = *reinterpret_cast<const uint4*>(w + ((int64_t)n * 9 + tap) * Cin + ch);
    *reinterpret_cast<uint4*>(out + idx * 8) = v;
}

int64_t g_hconv_launches = 0;
unsigned long long* g_hconv_trace = nullptr;  // (tests assert the route with it; not synchronised: a diagnostic)

// ---------------------------------------------------------------------------------------------------------------------------------
// Narrow form: N <= 16 output channels (the UNet's conv_out, 128 -> 8 at 4000 pixels; modeling_audioldm2.py:867).  The im2col kernels pad N to a
// 64-column tile (92.6 us for 2.4 GF of useful work); here the WEIGHTS are the stationary operand -- all 9 x Cin / 32 fragments of a 16-row MFMA A
// operand (rows >= N zero) live in each wave's registers for the whole launch -- and the pixels stream: the same halo tiles as above (three
// 40 KB buffers, two 64-channel chunks ahead, one barrier per chunk), B fragments of v_mfma_f32_16x16x32 (16 pixels x 32 channels) read with the
// tap's shift.  The kernel is bound by the halo stream (the level's activations once) -- the 16-row MFMAs at a quarter of their rows are 6 us of it.
struct HnP {
    const uint8_t* a;
    const uint8_t* wp;   // [Cin / 64][tap][2][64 lanes][8 elements]: A-operand fragments, lane = (row n = lane & 15, k-group lane >> 4)
    uint8_t* out;
    const uint8_t* bias;
    int64_t ldo;
    int32_t M, N, Cin, H, W, Wlog, Hs, Ws, Btot, m_tiles, nchunks;
    uint32_t a_bytes;
};

template <int DT> __device__ __forceinline__ f32x4 hn_mfma(u32x4 a, u32x4 b, f32x4 c) {
    if constexpr (DT == APAD_BF16) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

template <int DT, int NCH>  // NCH = Cin / 64 (1 or 2): 18 NCH weight fragments in registers
__global__ __launch_bounds__(512) void hnarrow_kernel(HnP p) {
    constexpr int NW = 8, NLD = 5, NPX = NLD * NW * 8, ABUF = NPX * 128, NBUF = 3, OFF_Z = NBUF * ABUF, BM = 256;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using E = ET<DT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int W = p.W, H = p.H, Wlog = p.Wlog, R = BM >> Wlog;
    const __amdgpu_buffer_rsrc_t ra = h_rsrc(p.a, p.a_bytes);
    const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr)smem;

    // the stationary operand
    u32x4 wf[NCH * 18];
#pragma unroll
    for (int i = 0; i < NCH * 18; ++i) wf[i] = *reinterpret_cast<const u32x4*>(p.wp + ((int64_t)i * 64 + lane) * 16);

    // halo DMA sources of a tile (as in hconv_kernel) and this lane's two pixels (16-pixel blocks 2 wave, 2 wave + 1 of the tile)
    auto tile_sources = [&](int mt, uint32_t (&ao)[5]) {
        const int g0 = mt * R, b0 = g0 / H, y0 = g0 - b0 * H;
#pragma unroll
        for (int l = 0; l < NLD; ++l) {
            const int pp = (l * NW + wave) * 8 + (lane >> 3), slot = lane & 7;
            const int j = pp >> Wlog, x = pp & (W - 1), v = y0 - 1 + j;
            const int q = v >= 0 ? v / (H + 1) : 0, r = v - q * (H + 1);
            const bool valid = v >= 0 && r != H && b0 + q < p.Btot;
            const int sy = p.Hs == H ? r : (r * p.Hs) / H, sx = p.Ws == W ? x : (x * p.Ws) / W;
            const uint32_t src = (uint32_t)(((b0 + q) * p.Hs + sy) * p.Ws + sx) * (uint32_t)(p.Cin * 2) + (uint32_t)((slot ^ ((pp >> 1) & 7)) << 4);
            ao[l] = valid ? src : H_OOB;
        }
    };
    auto pixel_records = [&](int mt, uint32_t (&pb)[2]) {
        const int g0 = mt * R, b0 = g0 / H, y0 = g0 - b0 * H;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ml = wave * 32 + b * 16 + l15, ir = ml >> Wlog, x = ml & (W - 1);
            pb[b] = (uint32_t)(((1 + ir + (y0 + ir) / H) << Wlog) + x);
        }
    };
    const bool x_first = (l15 & (W - 1)) == 0, x_last = (l15 & (W - 1)) == W - 1;  // (16 % W == 0: the same for both blocks)
    const uint32_t zbase = lds0 + (uint32_t)OFF_Z;  // 256 bytes of zeros: a border lane reads the zero slot of its own bank
    if (tid < 16) *reinterpret_cast<uint4*>(smem + OFF_Z + tid * 16) = make_uint4(0, 0, 0, 0);

    // the stream of (tile, chunk) pairs of this workgroup: chunk counter gc -> buffer gc % 3, requested two chunks ahead
    const int ntiles = p.m_tiles, tstride = (int)gridDim.x;
    uint32_t ao_cur[5], ao_next[5];
    auto request = [&](const uint32_t (&ao)[5], int chunk, int buf) {
#pragma unroll
        for (int l = 0; l < NLD; ++l)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(smem + buf * ABUF + (l * NW + wave) * 1024), 16, ao[l], chunk * 128, 0, 0);
    };
    int tl = blockIdx.x;
    tile_sources(tl, ao_cur);
    if (tl + tstride < ntiles) tile_sources(tl + tstride, ao_next);
    // prologue: the first two chunks of the stream
    int gc = 0;
    request(ao_cur, 0, 0);
    if (NCH == 2) request(ao_cur, 1, 1);
    else if (tl + tstride < ntiles) request(ao_next, 0, 1);
#pragma unroll 1
    for (; tl < ntiles; tl += tstride) {
        uint32_t pb[2];
        pixel_records(tl, pb);
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const bool more = tl + tstride < ntiles, more2 = tl + 2 * tstride < ntiles;
#pragma unroll
        for (int c = 0; c < NCH; ++c, ++gc) {
            const int buf = gc % NBUF;
            // chunk gc has landed: only the pieces of chunk gc + 1 (if the stream has one) may still be outstanding
            const bool has_next = c + 1 < NCH || more;
            if (has_next) h_wait_vm<NLD>();
            else h_wait_vm<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            H_BARRIER();  // every wave's pieces of chunk gc are in LDS, and every wave is past its reads of chunk gc - 1: that buffer takes chunk gc + 2
            {
                const int c2 = c + 2;  // chunk gc + 2 of the stream: of this tile, of the next one, or of the one after (NCH == 1)
                const int nbuf = (gc + 2) % NBUF;
                if (c2 < NCH) request(ao_cur, c2, nbuf);
                else if (c2 - NCH < NCH) { if (more) request(ao_next, c2 - NCH, nbuf); }
                else if (more2) {  // (NCH == 1: two tiles ahead -- its sources are computed here, the registers of ao_cur are free after the request below)
                    uint32_t ao2[5];
                    tile_sources(tl + 2 * tstride, ao2);
                    request(ao2, 0, nbuf);
                }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                uint32_t a0[2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const uint32_t ps = pb[b] + (uint32_t)(dy * W + dx);
                    uint32_t a = lds0 + (uint32_t)(buf * ABUF) + (ps << 7) + ((((ps >> 1) & 7) ^ (uint32_t)kg) << 4);
                    if (dx < 0) a = x_first ? (zbase | (a & 0xF0u)) : a;
                    if (dx > 0) a = x_last ? (zbase | (a & 0xF0u)) : a;
                    a0[b] = a;
                }
                u32x4 fb[2][2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int b = 0; b < 2; ++b) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[kk][b]) : "v"(a0[b] ^ (uint32_t)(kk << 6)));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[b] = hn_mfma<DT>(wf[(c * 9 + tap) * 2 + kk], fb[kk][b], acc[b]);
            }
        }
        // epilogue: lane (pixel l15 of block b, rows 4 kg .. 4 kg + 3): + bias -> storage type -> 8 bytes
        if (4 * kg < p.N) {
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = ld_elem<DT>(p.bias, 4 * kg + e);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int m = tl * BM + wave * 32 + b * 16 + l15;
                typename E::v4 h4;
#pragma unroll
                for (int e = 0; e < 4; ++e) h4[e] = (typename E::elem)(acc[b][e] + bv[e]);
                if (m < p.M) *reinterpret_cast<uint2*>(p.out + ((int64_t)m * p.ldo + 4 * kg) * 2) = __builtin_bit_cast(uint2, h4);
            }
        }
        if (more) {
#pragma unroll
            for (int l = 0; l < NLD; ++l) ao_cur[l] = ao_next[l];
            if (more2) tile_sources(tl + 2 * tstride, ao_next);
        }
    }
}

// w [N][tap][Cin] -> the narrow form's A-operand fragments (rows >= N zero)
template <typename TE>
__global__ void hnarrow_pack_kernel(const TE* __restrict__ w, TE* __restrict__ out, int N, int Cin, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one lane's 16 bytes
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int f = (int)(idx >> 6), kk = f & 1, tap = (f >> 1) % 9, c = (f >> 1) / 9;
    const int n = lane & 15, ch = c * 64 + kk * 32 + 8 * (lane >> 4);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < N) v = *reinterpret_cast<const uint4*>(w + ((int64_t)n * 9 + tap) * Cin + ch);
    *reinterpret_cast<uint4*>(out + idx * 8) = v;
}

template <int DT, int NCH> int hn_launch(const HnP& p, hipStream_t s) {
    auto kern = hnarrow_kernel<DT, NCH>;
    constexpr int SMEM = 3 * 5 * 8 * 8 * 128 + 256;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), SMEM, &devs) != 0) return -1;
    ++g_hconv_launches;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cus <= 0) cus = 256;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.m_tiles < cus ? p.m_tiles : cus)), dim3(512), SMEM, s, p);
    return apad_check_launch("apad_gemm(halo convolution, narrow form)");
}

// out = ((sum over the K slices, in slice order) + bias) + time-embedding row -> storage type -> + residual: the epilogue of a split convolution.
// Thread = 8 consecutive channels of one pixel.
template <int DT>
__global__ __launch_bounds__(256) void hconv_reduce_kernel(HcP p) {
    const int vpr = p.N >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)p.M * vpr) return;
    const int m = (int)(idx / vpr), n = (int)(idx - (int64_t)m * vpr) * 8;
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    for (int ks = 0; ks < p.ksplit; ++ks) {
        const float* src = p.partial + ((int64_t)ks * p.M + m) * p.N + n;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] += v0[e];
            a[e + 4] += v1[e];
        }
    }
    if (p.bias) {
        float b[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(p.bias + n * 2), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += b[e];
    }
    if (p.rg) {
        const int64_t step = p.step_ptr ? (int64_t)*p.step_ptr : 0;
        const int64_t grp = (p.rows_per_group >= p.M ? 0 : m / p.rows_per_group) + step;
        float t[8];
        unpack8<DT>(*reinterpret_cast<const uint4*>(p.rg + (grp * p.ld_rg + n) * 2), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += t[e];
    }
    uint4 o = pack8<DT>(a);
    if (p.residual) {
        float f[8], rr[8];
        unpack8<DT>(o, f);
        unpack8<DT>(*reinterpret_cast<const uint4*>(p.residual + ((int64_t)m * p.ldr + n) * 2), rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += rr[e];
        o = pack8<DT>(f);
    }
    *reinterpret_cast<uint4*>(p.out + ((int64_t)m * p.ldo + n) * 2) = o;
}

template <int DT, class T> int hc_launch(const HcP& p, hipStream_t s) {
    auto kern = hconv_kernel<DT, T>;
    ++g_hconv_launches;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), T::SMEM, &devs) != 0) return -1;
    // persistent workgroups, one per CU (the LDS footprint admits one): tile tl, tl + grid, ...
    static int cus[32] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    static const int grid_cap = [] { const char* e = getenv("APAD_HCONV_GRID"); return e ? atoi(e) : 0; }();  // A/B knob: workgroups (0 = one per CU)
    const int tiles = p.m_tiles * p.n_tiles * p.ksplit, cap = grid_cap > 0 ? grid_cap : cus[dev];
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(T::NT), T::SMEM, s, p);
    int rc = apad_check_launch("apad_gemm(halo convolution)");
    if (rc || p.ksplit == 1) return rc;
    const int64_t vecs = (int64_t)p.M * (p.N / 8);
    hipLaunchKernelGGL(hconv_reduce_kernel<DT>, dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, s, p);
    return apad_check_launch("apad_gemm(halo convolution, slice sum)");
}

using HcA = HcT<2, 4, 4, 2, 2>;  // 256 x 256, waves 128 x 64   (N % 256 == 0: the 1000-pixel level)
using HcB = HcT<4, 2, 2, 2, 4>;  // 256 x 128, waves  64 x 64   (N % 128 == 0: the 4000-pixel level)

}  // namespace

// K slices of a layer (a function of the LAYER only -- image width and input channels --, never of the row count): the 2-wide images of the
// 64-token level have 4 samples per 256-pixel tile, so a CFG batch of 64 is 16 row tiles x 5 column tiles = 80 workgroups of 90 .. 180 stages; three
// slices of whole 64-channel chunks make it 240 workgroups, the slabs are summed in slice order (hconv_reduce_kernel)
static int hconv_ksplit(int W, int Cin) { return W == 2 && Cin / 64 >= 3 ? 3 : 1; }

extern "C" int64_t apad_conv_halo_workspace_bytes(int64_t M, int64_t N, int64_t Cin, int32_t Wout) {
    const int ks = hconv_ksplit(Wout, (int)Cin);
    return ks > 1 ? ks * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int64_t apad_hconv_launch_count(void) { return g_hconv_launches; }
// probe builds (-DHC_TRACE=1) only: device buffer of 32 x workgroups uint64 time stamps; not part of the ABI header
extern "C" void apad_hconv_set_trace(void* buf) { g_hconv_trace = (unsigned long long*)buf; }

// bytes of the packed form: the wide form re-lays the N * 9 * Cin elements; the narrow form (N <= 16) is 18 fragments of 1 KB per 64-channel chunk
extern "C" int64_t apad_conv_halo_packed_bytes(int64_t N, int64_t Cin) { return N <= 16 ? (Cin / 64) * 18 * 1024 : N * 9 * Cin * 2; }

extern "C" int apad_conv_halo_pack(const void* w, void* out, int64_t N, int64_t Cin, int32_t dtype, void* stream) {
    APAD_CHECK(w && out && N > 0 && Cin > 0 && Cin % 64 == 0, "apad_conv_halo_pack: needs Cin %% 64 == 0");
    APAD_CHECK(dtype == APAD_BF16 || dtype == APAD_F16, "apad_conv_halo_pack: 16-bit weights only");
    if (N <= 16) {  // the narrow form: A-operand fragments of the 16-row MFMA, rows >= N zero
        const int64_t total = (Cin / 64) * 18 * 64;
        hipLaunchKernelGGL(hnarrow_pack_kernel<uint16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const uint16_t*)w, (uint16_t*)out, (int)N, (int)Cin, total);
        return apad_check_launch("apad_conv_halo_pack(narrow)");
    }
    const int64_t total = N * 9 * Cin / 8;
    hipLaunchKernelGGL(hconv_pack_kernel<uint16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)w, (uint16_t*)out, (int)N, (int)Cin, total);
    return apad_check_launch("apad_conv_halo_pack");
}

// Called by apad_gemm before its other dispatches.  1 = outside this kernel's envelope (the caller goes on), 0 = launched, < 0 = error.
// The envelope depends on the layer (geometry, channels, the packed weight form), never on the row count.
int apad_hconv_try(const apad_gemm_desc* d, hipStream_t s) {
    static const int mode = [] { const char* e = getenv("APAD_HCONV"); return e ? atoi(e) : 1; }();  // A/B knob: 0 = off
    if (!mode || !d->w_halo || d->a_mode != APAD_A_CONV3X3) return 1;
    if (d->dtype != APAD_BF16 && d->dtype != APAD_F16) return 1;
    if (d->epilogue != APAD_EPI_NONE || d->out_mode != APAD_OUT_ROWMAJOR || d->rowstat_out || d->rowstat_in) return 1;
    if (d->stride != 1 || d->src_batch_mod != 0 || d->conv_asym_pad || d->residual_row_mod != 0 || d->Cin % 64 != 0 || d->K != 9 * (int64_t)d->Cin) return 1;
    const int H = d->Hout, W = d->Wout;
    if (W < 2 || W > 16 || (W & (W - 1)) != 0 || H < 1) return 1;
    const int ksplit = hconv_ksplit(W, d->Cin);
    if (ksplit > 1 && (!d->workspace || d->workspace_bytes < ksplit * d->M * d->N * (int64_t)sizeof(float))) return 1;  // (the caller sizes it with
                                                                                                                       //  apad_conv_halo_workspace_bytes)
    if (d->Hup == 0 && (d->Hin != H || d->Win != W)) return 1;
    if (d->Hup != 0 && (d->Hup != H || d->Wup != W)) return 1;
    if (d->M % ((int64_t)H * W) != 0 || d->M >= (1LL << 30)) return 1;
    if (d->N <= 16) {  // the narrow form (conv_out): weights stationary in registers
        if (d->N % 8 != 0 || (d->Cin != 64 && d->Cin != 128) || W < 4 || d->residual || d->rowgroup_bias || d->ldo % 8 != 0) return 1;
        const int R = 256 / W, S = (R - 1) / H + 1;
        if ((R + 2 + S) * W > 320) return 1;
        const int64_t Btot = d->M / ((int64_t)H * W);
        const int64_t a_bytes = Btot * d->Hin * d->Win * (int64_t)d->Cin * 2;
        if (a_bytes >= (1LL << 31)) return 1;
        HnP q;
        q.a = (const uint8_t*)d->a; q.wp = (const uint8_t*)d->w_halo; q.out = (uint8_t*)d->out; q.bias = (const uint8_t*)d->bias; q.ldo = d->ldo;
        q.M = (int32_t)d->M; q.N = (int32_t)d->N; q.Cin = d->Cin; q.H = H; q.W = W; q.Wlog = __builtin_ctz((unsigned)W); q.Hs = d->Hin; q.Ws = d->Win;
        q.Btot = (int32_t)Btot; q.m_tiles = (int32_t)((d->M + 255) / 256); q.nchunks = d->Cin / 64; q.a_bytes = (uint32_t)a_bytes;
        if (d->dtype == APAD_BF16) return d->Cin == 64 ? hn_launch<APAD_BF16, 1>(q, s) : hn_launch<APAD_BF16, 2>(q, s);
        return d->Cin == 64 ? hn_launch<APAD_F16, 1>(q, s) : hn_launch<APAD_F16, 2>(q, s);
    }
    if (d->N % 128 != 0) return 1;
    if (d->ldo % 8 != 0 || (d->residual && d->ldr % 8 != 0)) return 1;
    // tile shape by the LAYER (never the row count): 256 x 256 when N % 256 == 0, else 256 x 128.  (A four-wave 128 x 192 shape for N = 384 --
    // 252 instead of 189 workgroups at batch 32 -- measured slower, 51.8 vs 48.2 us: one wave per SIMD does not cover its own barrier bubbles.)
    const int cfg = d->N % 256 == 0 ? 0 : 1;
    constexpr int BM = 256;
    const int BNc = cfg == 0 ? HcA::BN : HcB::BN;
    const int R = BM / W, S = (R - 1) / H + 1;
    if ((R + 2 + S) * W > HcA::NPX) return 1;
    const int64_t Btot = d->M / ((int64_t)H * W);
    const int64_t a_bytes = Btot * d->Hin * d->Win * (int64_t)d->Cin * 2, w_bytes = d->N * 9 * (int64_t)d->Cin * 2;
    if (a_bytes >= (1LL << 31) || w_bytes >= (1LL << 31)) return 1;
    if (d->rowgroup_bias && d->rows_per_group <= 0) return 1;
    HcP p;
    p.a = (const uint8_t*)d->a; p.wp = (const uint8_t*)d->w_halo; p.out = (uint8_t*)d->out; p.bias = (const uint8_t*)d->bias;
    p.residual = (const uint8_t*)d->residual; p.rg = (const uint8_t*)d->rowgroup_bias; p.step_ptr = d->step_ptr;
    p.ldo = d->ldo; p.ldr = d->ldr; p.ld_rg = d->ld_rg; p.rows_per_group = d->rows_per_group;
    p.M = (int32_t)d->M; p.N = (int32_t)d->N; p.Cin = d->Cin;
    p.H = H; p.W = W; p.Wlog = __builtin_ctz((unsigned)W); p.Hs = d->Hin; p.Ws = d->Win; p.Btot = (int32_t)Btot;
    p.m_tiles = (int32_t)((d->M + BM - 1) / BM); p.n_tiles = (int32_t)(d->N / BNc); p.nchunks = d->Cin / 64;
    p.ksplit = ksplit; p.partial = (float*)d->workspace;
    p.a_bytes = (uint32_t)a_bytes; p.w_bytes = (uint32_t)w_bytes; p.trace = g_hconv_trace;
    if (d->dtype == APAD_BF16) return cfg == 0 ? hc_launch<APAD_BF16, HcA>(p, s) : hc_launch<APAD_BF16, HcB>(p, s);
    return cfg == 0 ? hc_launch<APAD_F16, HcA>(p, s) : hc_launch<APAD_F16, HcB>(p, s);
}
