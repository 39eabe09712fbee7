// Big-tile GEMM / implicit 3x3 convolution for the compute-bound launches of the path (the resnet convolutions of the 4000- and
// 1000-pixel UNet levels: M = 64 000 .. 256 000 output pixels, K = 9 Cin = 1152 .. 5760, N = 128 / 256; modeling_audioldm2.py's
// ResnetBlock2D conv1 / conv2), selected by apad_gemm (gemm.hip) when the problem fits -- same descriptor, same results.
//
// What differs from the 128x128 tiled kernel (gemm.hip), whose k-loop spends as many issue cycles on the gather's address
// arithmetic, its exec-masked loads and the register -> LDS copy as on its MFMAs:
//   * 512 threads = 8 waves on a 256 x 128 x 64 tile (4 x 2 waves of 64 x 64, three 48 KB LDS stages) or, when N % 256 == 0, a
//     256 x 256 x 64 tile (2 x 4 waves of 128 x 64, two 64 KB stages): measured with ablation builds (CG_ABL), the 256 x 128 form is
//     bound by the CU's LDS fill + fragment-read traffic (48 KB in, 128 KB out per k-tile against 1024 MFMA cycles), not by the MFMAs
//     -- the square tile moves a third fewer bytes per FLOP through L2 -> LDS and a quarter fewer through LDS -> registers
//   * operands go HBM / L2 -> LDS directly (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write pass.  The DMA writes
//     lane-linear, so the bank-conflict-free XOR layout of the tile is produced on the SOURCE side (lane -> (row, 16-byte chunk)).
//     The convolution's zero padding is the buffer range check: a lane whose filter tap falls outside the image gets an
//     out-of-range offset and the hardware writes zeros (tools/probes/buflds.hip) -- no branches, no selects on data
//   * a tile is requested NST - 1 k-tiles ahead and waited for with a COUNTED s_waitcnt vmcnt, raw s_barrier (a __syncthreads()
//     would drain the queue), so loads stay in flight across barriers; the DMA instructions are issued between the MFMAs
//   * the two waves of a SIMD (waves w and w + 4) run half a phase apart: a phase is [fragment reads | barrier | 8 MFMAs + DMA issue |
//     barrier]; while one wave of the SIMD is in its MFMA segment the other is in its read segment
//     (MI355X_MICROARCH.md, "Two waves per SIMD")
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

#ifndef CG_ABL
#define CG_ABL 0  // ablation bits for timing-only probe builds (tools/ab_build.sh): 1 no DMA in the loop, 2 no fragment reads, 4 no MFMAs
#endif
constexpr int CBM = 256, CBK = 64;
constexpr int CA_BYTES = CBM * CBK * 2;       // 32 768
constexpr uint32_t C_OOB = 0x80000000u;       // an offset no operand reaches (sizes are checked < 2 GB on the host)

// BN = 128: 4 x 2 waves of 64 x 64 (MI = 2 MFMA tiles down), phases of two k-steps, 3 stages.  BN = 256: 2 x 4 waves of 128 x 64
// (MI = 4), phases of one k-step, 2 stages.  Either way a phase is 8 MFMAs of 32x32x16 per wave.
template <int BN> struct CgT {
    static constexpr int MI = BN == 128 ? 2 : 4;         // 32-row MFMA tiles per wave
    static constexpr int WAVES_N = BN / 64;              // 64 columns per wave
    static constexpr int KSP = BN == 128 ? 2 : 1;        // k-steps (of 16) per phase
    static constexpr int NPH = 4 / KSP;                  // phases per k-tile
    static constexpr int NST = BN == 128 ? 3 : 2;        // LDS stages
    static constexpr int D = NST - 1;                    // a tile is requested D k-tiles ahead
    static constexpr int B_BYTES = BN * CBK * 2;
    static constexpr int STAGE = CA_BYTES + B_BYTES;     // 49 152 / 65 536
    static constexpr int PB = BN / 64;                   // B pieces (1 KB DMA instructions) per wave and k-tile; A: 4
    static constexpr int P = 4 + PB;                     // 6 / 8
    static constexpr int C_LD = BN + 8;                  // epilogue tile row stride (elements); the tile holds 128 rows
    static constexpr int SMEM = NST * STAGE;             // 147 456 / 131 072  (>= 128 * C_LD * 2)
    static_assert(P <= 3 * (NST == 2 ? NPH - 1 : NPH), "three DMA pieces per phase; none in the last phase of a two-stage pipeline");
};

struct CgP {
    const uint8_t* a;
    const uint8_t* w;
    uint8_t* out;
    const uint8_t* bias;
    const uint8_t* residual;
    const uint8_t* rg;
    const int32_t* step_ptr;
    int64_t ldo, ldr, ld_rg;
    int32_t M, N, K, lda, ldw;
    int32_t Hin, Win, Cin, res_mod;
    int32_t Hout, Wout, stride, Hup, Wup;  // general convolution form (stride 2 / nearest-upsampled source)
    int32_t m_tiles, n_tiles;
    uint32_t a_bytes, w_bytes;
    // two-source plain A (apad_gemm_desc::a2): k-tiles from ksplit on come from a2
    const uint8_t* a2;
    int32_t lda2, ksplit, a_mod, a2_mod;
    uint32_t a2_bytes;
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t c_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// General 3x3-convolution source mapping (stride 2: Downsample2D; nearest-upsampled source: Upsample2D with the interpolation folded
// into the gather, modeling_audioldm2.py:1156 / :1509): the source pixel of tap (ky, kx) is not "centre + a uniform delta" any more,
// so a row keeps three row offsets yo (byte offset of (b * Hin + sy(ky)) * Win * Cin * 2), three column offsets xo (sx(kx) * Cin * 2 +
// chunk * 16) and validity bits ok (bit ky: row tap inside the (virtual) source grid; bit 3 + kx: column tap inside; 0 for rows past
// M); a k-tile picks one of each (ky, kx are wave-uniform).  (Plain scalars / arrays with literal indices: a struct of arrays handed
// to the lambdas by reference lived in scratch.)
#define CG_ROW_GENERAL(yo, xo, ok, p, m, c)                                                                          \
    do {                                                                                                              \
        const int hw_ = (p).Hout * (p).Wout;                                                                          \
        const int b_ = (m) / hw_, rem_ = (m) - b_ * hw_;                                                              \
        const int oy_ = rem_ / (p).Wout, ox_ = rem_ - oy_ * (p).Wout;                                                 \
        const int Hs_ = (p).Hup > 0 ? (p).Hup : (p).Hin, Ws_ = (p).Hup > 0 ? (p).Wup : (p).Win;                       \
        (ok) = 0;                                                                                                     \
        _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) {                                                            \
            const int uy_ = oy_ * (p).stride + k_ - 1, ux_ = ox_ * (p).stride + k_ - 1;                               \
            const bool vy_ = (unsigned)uy_ < (unsigned)Hs_, vx_ = (unsigned)ux_ < (unsigned)Ws_;                      \
            const int sy_ = vy_ ? ((p).Hup > 0 ? (int)(((int64_t)uy_ * (p).Hin) / (p).Hup) : uy_) : 0;                \
            const int sx_ = vx_ ? ((p).Hup > 0 ? (int)(((int64_t)ux_ * (p).Win) / (p).Wup) : ux_) : 0;                \
            (yo)[k_] = (uint32_t)((b_ * (p).Hin + sy_) * (p).Win * (p).Cin * 2);                                      \
            (xo)[k_] = (uint32_t)(sx_ * (p).Cin * 2 + (c) * 16);                                                      \
            if ((m) < (p).M) (ok) |= (vy_ ? 1u << k_ : 0u) | (vx_ ? 8u << k_ : 0u);                                   \
        }                                                                                                             \
    } while (0)
__device__ __forceinline__ uint32_t cg_pick3(const uint32_t (&v)[3], int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }

#define C_FENCE() asm volatile("" ::: "memory")
#define C_BARRIER()                          \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        C_FENCE();                           \
        __builtin_amdgcn_s_barrier();        \
        C_FENCE();                           \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)

template <int N_> __device__ __forceinline__ void c_wait_vm() {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N_ == 0, "add the count");
}

template <int DT, int MODE, int BN>  // MODE: 0 plain A, 1 3x3 convolution (stride 1), 2 general 3x3 convolution (stride 2 / up-sampled source)
__global__ __launch_bounds__(512) void cgemm_kernel(CgP p) {
    constexpr bool CONV = MODE != 0, GEN = MODE == 2;
    using T = CgT<BN>;
    constexpr int MI = T::MI, KSP = T::KSP, NPH = T::NPH, NST = T::NST, D = T::D, PB = T::PB, P = T::P, STAGE = T::STAGE, C_LD = T::C_LD;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using E = ET<DT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;  // waves w and w + 4 share a SIMD: the second half of the workgroup runs one barrier behind
    const int wm = wave / T::WAVES_N, wn = wave % T::WAVES_N;
    const int half = lane >> 5, l31 = lane & 31;

    // XCD-aware tile order (speed only): all N-tiles of one M-tile share blockIdx % 8, i.e. one XCD's L2 fetches an A panel once
    int mt, nt;
    {
        const int nN = p.n_tiles, nM = p.m_tiles;
        const int b = blockIdx.x;
        const int full = (nM / 8) * 8 * nN;
        if (b < full) {
            const int g = b / (8 * nN), rem = b - g * 8 * nN;
            nt = rem >> 3;
            mt = g * 8 + (rem & 7);
        } else {
            const int rem = b - full, tail = nM - (nM / 8) * 8;
            nt = rem / tail;
            mt = (nM / 8) * 8 + rem - nt * tail;
        }
    }
    const int m0 = mt * CBM, n0 = nt * BN;

    // ---- DMA sources.  One instruction of a wave fills one 1 KB block = 8 tile rows x 128 bytes; lane -> (row r = lane / 8,
    //      LDS slot lane % 8), and the slot holds source chunk slot ^ ((row >> 1) & 7): the swizzle the fragment reads undo. ----
    const __amdgpu_buffer_rsrc_t ra = c_rsrc(p.a, p.a_bytes), rw = c_rsrc(p.w, p.w_bytes);
    uint32_t aoff2[4];  // plain, two sources: the row's offset in the second one
    uint32_t aoff[4];   // plain: byte offset of (row, chunk) at k = 0, or C_OOB; conv: of the CENTRE tap, channel 0
    uint32_t amask[4];  // conv: bit (3 ky + kx) = that tap lies inside the image (0 for rows past M)
    uint32_t gyo[4][3], gxo[4][3], gok[4];  // (GEN only)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int R = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        const int m = m0 + R;
        const bool valid = m < p.M;
        amask[i] = 0;
        if constexpr (GEN) {
            CG_ROW_GENERAL(gyo[i], gxo[i], gok[i], p, m, c);
            aoff[i] = 0;
        } else if (CONV) {
            const int hw = p.Hin * p.Win;
            const int b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Win, ox = rem - oy * p.Win;
            aoff[i] = (uint32_t)(((b * p.Hin + oy) * p.Win + ox) * p.Cin * 2 + c * 16);
            if (valid) {
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
                    if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) amask[i] |= 1u << t;
                }
            }
        } else {
            aoff[i] = valid ? (uint32_t)((p.a_mod > 0 ? m % p.a_mod : m) * p.lda * 2 + c * 16) : C_OOB;
            aoff2[i] = (valid && p.a2 != nullptr) ? (uint32_t)((p.a2_mod > 0 ? m % p.a2_mod : m) * p.lda2 * 2 + c * 16) : C_OOB;
        }
    }
    uint32_t boff[4];  // (PB used; a template-dependent array bound captured by the lambdas below loses the kernel's host stub: hipcc 7.2)
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int R = (wave * PB + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        boff[j] = (uint32_t)((n0 + R) * p.ldw * 2 + c * 16);
    }
    const int nk = p.K / CBK;
    const int tiles_per_tap = CONV ? p.Cin / CBK : 1;
    // (tap, channel block) of the next tile to REQUEST: tiles are requested in order, one per iteration
    int rq_tap = 0, rq_cb = 0;
    uint32_t av[4];
    int a_soff = 0;
    bool a_second = false;  // (wave-uniform) the tile being requested lies in the second source
    auto next_tile_sources = [&](int kt) {  // per-lane A offsets + the scalar offset of k-tile kt
        if (CONV) {
            const int ky = rq_tap / 3, kx = rq_tap - ky * 3;
            if constexpr (GEN) {
                const uint32_t need = (1u << ky) | (8u << kx);
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = (gok[i] & need) == need ? cg_pick3(gyo[i], ky) + cg_pick3(gxo[i], kx) : C_OOB;
            } else {
                const int delta = ((ky - 1) * p.Win + (kx - 1)) * p.Cin * 2;  // wave-uniform, may be negative
                const uint32_t bit = 1u << rq_tap;
#pragma unroll
                for (int i = 0; i < 4; ++i) av[i] = (amask[i] & bit) ? aoff[i] + (uint32_t)delta : C_OOB;
            }
            a_soff = rq_cb * (CBK * 2);
            if (++rq_cb == tiles_per_tap) {
                rq_cb = 0;
                ++rq_tap;
            }
        } else {
            a_second = p.a2 != nullptr && kt * CBK >= p.ksplit;
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = a_second ? aoff2[i] : aoff[i];
            a_soff = (a_second ? kt * CBK - p.ksplit : kt * CBK) * 2;
        }
    };
    // piece q of k-tile kt (its sources prepared by next_tile_sources) -> LDS stage `stage`: q < 4 an A block, else a B block
    auto issue = [&](int q, int stage, int kt) {
        if (q < 4) {
            if constexpr (CONV) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(smem + stage * STAGE + (wave * 4 + q) * 1024), 16, av[q], a_soff, 0, 0);
            } else {
                // (the descriptor is rebuilt from scalar selects: two descriptors selected per call were kept in scratch)
                const __amdgpu_buffer_rsrc_t rs = c_rsrc(a_second ? p.a2 : p.a, a_second ? p.a2_bytes : p.a_bytes);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + stage * STAGE + (wave * 4 + q) * 1024), 16, av[q], a_soff, 0, 0);
            }
        } else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + stage * STAGE + CA_BYTES + (wave * PB + q - 4) * 1024), 16,
                                                     boff[q - 4], kt * (CBK * 2), 0, 0);
    };

    // ---- fragment addresses: row (base + l31), chunk ks*2 + half -> row*128 + ((chunk ^ ((row >> 1) & 7)) << 4); the row bases are
    //      multiples of 32, so the swizzle term depends on l31 only ----
    uint32_t fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = (uint32_t)(l31 * 128 + (((ks * 2 + half) ^ ((l31 >> 1) & 7)) << 4));
    const uint32_t abase = (uint32_t)(wm * MI * 32 * 128), bbase = (uint32_t)(CA_BYTES + wn * 64 * 128);

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment reads are inline asm: the compiler's wait-count pass orders every ds_read it can see behind ALL outstanding LDS-DMA
    // (it inserts s_waitcnt vmcnt(0) in front of the first read of each phase, which drains the tiles in flight); the ordering
    // that is actually needed -- the tile being read was waited for with the counted vmcnt below, by every wave, one barrier ago --
    // is kept by hand.  The values are consumed behind an explicit lgkmcnt(0) + sched_barrier (the pass does not see the reads).
    u32x4 fa[KSP][MI] = {}, fb[KSP][2] = {};  // [k-step of the phase][MFMA tile]
    const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr)smem;
    auto read_frags = [&](int stage, int ph) {
        if (CG_ABL & 2) return;
        const uint32_t st = lds0 + (uint32_t)(stage * STAGE);
#pragma unroll
        for (int u = 0; u < KSP; ++u) {
            const uint32_t aa = st + abase + fo[ph * KSP + u], bb = st + bbase + fo[ph * KSP + u];
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[u][0]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[u][1]) : "v"(aa));
            if constexpr (MI == 4) {
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fa[u][2]) : "v"(aa));
                asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(fa[u][3]) : "v"(aa));
            }
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[u][0]) : "v"(bb));
            asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[u][1]) : "v"(bb));
        }
    };
    // the MFMA segment of a phase; `dma(u)` (u = 0..2) issues this wave's next DMA pieces BETWEEN the MFMAs (an LDS-DMA instruction
    // costs ~60 issue cycles among bare MFMAs and 100+ in a segment that also carries the fragment reads: MI355X_MICROARCH.md)
    auto mfmas = [&](auto&& dma) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        int n = 0;
#pragma unroll
        for (int u = 0; u < KSP; ++u)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (!(CG_ABL & 4))
                        acc[i][j] = E::mfma32(__builtin_bit_cast(typename E::v8, fa[u][i]), __builtin_bit_cast(typename E::v8, fb[u][j]), acc[i][j]);
                    if (n < 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        dma(n);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    ++n;
                }
        __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: the first D tiles requested; tile 0 waited for ----
#pragma unroll
    for (int t = 0; t < D; ++t)
        if (t < nk) {
            next_tile_sources(t);
#pragma unroll
            for (int q = 0; q < P; ++q) issue(q, t, t);
        }
    if (D > 1 && nk > 1) c_wait_vm<(D > 1 ? P : 0)>();  // (D == 2: the second tile's pieces may stay in flight)
    else c_wait_vm<0>();
    C_BARRIER();
    if (grp == 1) C_BARRIER();  // the stagger

    // One k-tile.  Stage indices are compile-time (the loop is unrolled by the stage count).
    auto ktile = [&](int t, auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value, SR = (S + D) % NST;  // SR: the stage tile t + D goes to (it held tile t - 1)
        const bool req = (CG_ABL & 1) ? false : t + D < nk;  // wave-uniform
#pragma unroll
        for (int ph = 0; ph < NPH; ++ph) {
            read_frags(S, ph);
            if (ph == 0 && req) next_tile_sources(t + D);
            if (ph == NPH - 1) {
                // tile t + 1 must have landed before anyone reads it behind the next barriers; with three stages the pieces of
                // tile t + 2 issued in this tile's earlier phases may stay in flight
                constexpr int inflight = NST == 3 ? (3 * (NPH - 1) < P ? 3 * (NPH - 1) : P) : 0;
                if (NST == 3 && req) c_wait_vm<inflight>();
                else c_wait_vm<0>();
                // the last reads of stage S are complete before the barrier behind which the other half may overwrite it
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            C_BARRIER();
            mfmas([&](int u) {
                const int q = ph * 3 + u;
                if (q < P && req) issue(q, SR, t + D);
            });
            C_BARRIER();
        }
    };
    if constexpr (NST == 3) {
#pragma unroll 1
        for (int t = 0; t < nk; t += 3) {
            ktile(t, std::integral_constant<int, 0>{});
            if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
            if (t + 2 < nk) ktile(t + 2, std::integral_constant<int, 2>{});
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < nk; t += 2) {
            ktile(t, std::integral_constant<int, 0>{});
            if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
        }
    }
    if (grp == 0) C_BARRIER();
    // (every wave is past its last fragment read and its last DMA wait: the stages are dead)

    // ---- epilogue, 128 tile rows at a time: acc + bias + time-embedding row -> storage type -> LDS tile -> + residual -> full-row
    //      16-byte stores ----
    typename E::elem* ct = reinterpret_cast<typename E::elem*>(smem);
    const int64_t step = p.step_ptr ? (int64_t)*p.step_ptr : 0;
    float bv[2], rg0[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l31;
        bv[j] = p.bias ? ld_elem<DT>(p.bias, n) : 0.f;
        rg0[j] = p.rg ? ld_elem<DT>(p.rg, step * p.ld_rg + n) : 0.f;
    }
    constexpr int VPR = BN / 8, NV = 128 * VPR / 512;  // 16-byte vectors per output row; vectors per thread and half
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        // the residual rows of this half are requested first: their latency runs under the accumulator -> LDS pass
        uint4 rres[NV];
        if (p.residual) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int idx = tid + v * 512, rl = idx / VPR, vc = idx - rl * VPR;
                int m = m0 + hh * 128 + rl;
                m = m < p.M ? m : p.M - 1;
                const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
                rres[v] = *reinterpret_cast<const uint4*>(p.residual + (rm * p.ldr + n0 + vc * 8) * 2);
            }
        }
        if ((wm * MI * 32) / 128 == hh) {  // this wave's rows lie in this half (wave-uniform)
            const int rbase = wm * MI * 32 - hh * 128;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nl = wn * 64 + j * 32 + l31;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        ct[ml * C_LD + nl] = (typename E::elem)(acc[i][j][r] + bv[j] + rg0[j]);
                    }
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * 512, rl = idx / VPR, vc = idx - rl * VPR;
            const int m = m0 + hh * 128 + rl, n = n0 + vc * 8;
            uint4 o = *reinterpret_cast<const uint4*>(&ct[rl * C_LD + vc * 8]);
            if (p.residual) {
                float f[8], rr[8];
                unpack8<DT>(o, f);
                unpack8<DT>(rres[v], rr);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += rr[e];
                o = pack8<DT>(f);
            }
            if (m < p.M) *reinterpret_cast<uint4*>(p.out + ((int64_t)m * p.ldo + n) * 2) = o;
        }
        if (hh == 0) __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Small-tile form for the 3x3 convolutions of the 64-token level (M = 2048 / 4096 output pixels, K = 9 Cin = 3456 .. 11520, N = 640)
// and of small batches: 64 x 64 x 64 tile, 4 waves (2 x 2 of 32 x 32), FOUR 16 KB LDS stages -- three k-tiles in flight per workgroup,
// two workgroups per CU.  Those launches are a few hundred workgroups walking 54 .. 180 k-tiles each: the tiled kernel keeps ONE
// k-tile in flight per workgroup (global load -> registers -> LDS -> barrier per k-tile, ~0.45 .. 0.75 us each), so its loop is a chain
// of exposed load latencies; here the DMA ring keeps the loads of three k-tiles outstanding and the loop is paced by the CU's LDS fill
// rate.  Same operand layout, zero padding through the buffer range check and sequential k order as the big-tile kernel above (a
// row's result does not depend on which of the two forms its batch size selects).  One barrier per k-tile: [counted vmcnt -> barrier
// -> DMA issue of tile t + 3 into the stage read in iteration t - 1 -> fragment reads -> 4 MFMAs].
constexpr int SBM = 64, SBN = 64, SSTAGE = 2 * SBM * CBK * 2, SNST = 4, SSMEM = SNST * SSTAGE;  // 16 384 per stage, 65 536

template <int DT, bool GEN>
__global__ __launch_bounds__(256) void cconv_small_kernel(CgP p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    using E = ET<DT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    int mt, nt;
    {
        const int nN = p.n_tiles, nM = p.m_tiles;
        const int b = blockIdx.x;
        const int full = (nM / 8) * 8 * nN;
        if (b < full) {
            const int g = b / (8 * nN), rem = b - g * 8 * nN;
            nt = rem >> 3;
            mt = g * 8 + (rem & 7);
        } else {
            const int rem = b - full, tail = nM - (nM / 8) * 8;
            nt = rem / tail;
            mt = (nM / 8) * 8 + rem - nt * tail;
        }
    }
    const int m0 = mt * SBM, n0 = nt * SBN;
    const __amdgpu_buffer_rsrc_t ra = c_rsrc(p.a, p.a_bytes), rw = c_rsrc(p.w, p.w_bytes);
    // DMA sources: wave w fills blocks 2w, 2w + 1 (8 rows each) of the A tile and of the W tile
    uint32_t aoff[2], amask[2], boff[2];
    uint32_t gyo[2][3], gxo[2][3], gok[2];  // (GEN only)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (wave * 2 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((R >> 1) & 7);
        const int m = m0 + R;
        if constexpr (GEN) CG_ROW_GENERAL(gyo[i], gxo[i], gok[i], p, m, c);
        const int hw = p.Hin * p.Win;
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.Win, ox = rem - oy * p.Win;
        aoff[i] = (uint32_t)(((b * p.Hin + oy) * p.Win + ox) * p.Cin * 2 + c * 16);
        amask[i] = 0;
        if (m < p.M) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
                if ((unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win) amask[i] |= 1u << t;
            }
        }
        boff[i] = (uint32_t)((n0 + R) * p.ldw * 2 + c * 16);
    }
    const int nk = p.K / CBK, tiles_per_tap = p.Cin / CBK;
    int rq_tap = 0, rq_cb = 0;
    auto request = [&](int kt, int stage) {  // the four pieces of this wave for k-tile kt
        const int ky = rq_tap / 3, kx = rq_tap - ky * 3;
        const int delta = ((ky - 1) * p.Win + (kx - 1)) * p.Cin * 2;
        const uint32_t bit = 1u << rq_tap;
        const int soff = rq_cb * (CBK * 2);
        if (++rq_cb == tiles_per_tap) {
            rq_cb = 0;
            ++rq_tap;
        }
        uint8_t* st = smem + stage * SSTAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t need = (1u << ky) | (8u << kx);
            const uint32_t av = GEN ? ((gok[i] & need) == need ? cg_pick3(gyo[i], ky) + cg_pick3(gxo[i], kx) : C_OOB)
                                    : ((amask[i] & bit) ? aoff[i] + (uint32_t)delta : C_OOB);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(st + (wave * 2 + i) * 1024), 16, av, soff, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(st + SBM * CBK * 2 + (wave * 2 + i) * 1024), 16, boff[i], kt * (CBK * 2), 0, 0);
    };
    uint32_t fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = (uint32_t)(l31 * 128 + (((ks * 2 + half) ^ ((l31 >> 1) & 7)) << 4));
    const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr)smem;
    const uint32_t abase = lds0 + (uint32_t)(wm * 32 * 128), bbase = lds0 + (uint32_t)(SBM * CBK * 2 + wn * 32 * 128);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

#pragma unroll
    for (int t = 0; t < SNST - 1; ++t)
        if (t < nk) request(t, t);
    auto ktile = [&](int t, auto stage_tag) {
        constexpr int S = decltype(stage_tag)::value;
        // this wave's pieces of tile t have landed when at most the pieces of the (up to two) later tiles are outstanding
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C_BARRIER();  // everyone's pieces of tile t are in LDS, and everyone is past its reads of tile t - 1
        if (t + SNST - 1 < nk) request(t + SNST - 1, (S + SNST - 1) % SNST);
        u32x4 fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t aa = abase + (uint32_t)(S * SSTAGE) + fo[ks], bb = bbase + (uint32_t)(S * SSTAGE) + fo[ks];
            asm volatile("ds_read_b128 %0, %1" : "=v"(fa[ks]) : "v"(aa));
            asm volatile("ds_read_b128 %0, %1" : "=v"(fb[ks]) : "v"(bb));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            acc = E::mfma32(__builtin_bit_cast(typename E::v8, fa[ks]), __builtin_bit_cast(typename E::v8, fb[ks]), acc);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int t = 0; t < nk; t += 4) {
        ktile(t, std::integral_constant<int, 0>{});
        if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
        if (t + 2 < nk) ktile(t + 2, std::integral_constant<int, 2>{});
        if (t + 3 < nk) ktile(t + 3, std::integral_constant<int, 3>{});
    }
    __syncthreads();  // the stages are dead

    constexpr int LD = SBN + 8;
    typename E::elem* ct = reinterpret_cast<typename E::elem*>(smem);
    const int64_t step = p.step_ptr ? (int64_t)*p.step_ptr : 0;
    {
        const int nl = wn * 32 + l31, n = n0 + nl;
        const float bv = p.bias ? ld_elem<DT>(p.bias, n) : 0.f;
        const float rg0 = p.rg ? ld_elem<DT>(p.rg, step * p.ld_rg + n) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            ct[ml * LD + nl] = (typename E::elem)(acc[r] + bv + rg0);
        }
    }
    __syncthreads();
    constexpr int VPR = SBN / 8;
#pragma unroll
    for (int v = 0; v < SBM * VPR / 256; ++v) {
        const int idx = tid + v * 256, rl = idx / VPR, vc = idx - rl * VPR;
        const int m = m0 + rl, n = n0 + vc * 8;
        if (m >= p.M) continue;
        uint4 o = *reinterpret_cast<const uint4*>(&ct[rl * LD + vc * 8]);
        if (p.residual) {
            float f[8], rr[8];
            unpack8<DT>(o, f);
            const int64_t rm = p.res_mod > 0 ? m % p.res_mod : m;
            unpack8<DT>(*reinterpret_cast<const uint4*>(p.residual + (rm * p.ldr + n) * 2), rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += rr[e];
            o = pack8<DT>(f);
        }
        *reinterpret_cast<uint4*>(p.out + ((int64_t)m * p.ldo + n) * 2) = o;
    }
}

template <int DT, bool GEN> int cs_launch(const CgP& p, hipStream_t s) {
    auto kern = cconv_small_kernel<DT, GEN>;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), SSMEM, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.m_tiles * p.n_tiles)), dim3(256), SSMEM, s, p);
    return apad_check_launch("apad_gemm(small-tile conv)");
}

template <int DT, int MODE, int BN> int cg_launch(const CgP& p, hipStream_t s) {
    auto kern = cgemm_kernel<DT, MODE, BN>;
    constexpr int CSMEM = CgT<BN>::SMEM;
    static unsigned devs = 0;
    if (apad_ensure_dyn_lds(reinterpret_cast<const void*>(kern), CSMEM, &devs) != 0) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.m_tiles * p.n_tiles)), dim3(512), CSMEM, s, p);
    return apad_check_launch("apad_gemm(big tile)");
}

}  // namespace

// Called by apad_gemm before its own dispatch.  Returns 1 when the problem is outside this kernel's envelope (the caller goes on),
// 0 after a launch, < 0 on a launch error.
int apad_cgemm_try(const apad_gemm_desc* d, hipStream_t s) {
    static const int mode = [] { const char* e = getenv("APAD_CGEMM"); return e ? atoi(e) : 1; }();  // A/B knob: 0 = off
    constexpr long min_rows = 16000L;  // (step-level A/B: 32768 -> 16000 = -1.0 ms, 4000 / 2000: no further gain)
    if (!mode) return 1;
    if (d->dtype != APAD_BF16 && d->dtype != APAD_F16) return 1;
    if (d->epilogue != APAD_EPI_NONE || d->out_mode != APAD_OUT_ROWMAJOR || d->rowstat_out || d->rowstat_in) return 1;
    constexpr int bn_mode = 0;  // A/B knob: 128 = never the square tile
    constexpr int small_mode = 1;  // A/B knob: 0 = off
    // below the row threshold: the small-tile form for 3x3 convolutions with long reductions (the 64-token level)
    const bool small = d->M < min_rows && small_mode && d->a_mode == APAD_A_CONV3X3 && d->K >= 2304 && d->N % SBN == 0;
    if ((d->N % 128 != 0 && !small) || d->K % CBK != 0 || (d->M < min_rows && !small) || d->M >= (1LL << 30)) return 1;
    const bool sq = d->N % 256 == 0 && bn_mode != 128;
    if (d->rowgroup_bias && d->rows_per_group < d->M) return 1;  // only the table form (every row reads row *step_ptr)
    const bool conv = d->a_mode == APAD_A_CONV3X3;
    bool general = false;
    int64_t a_bytes;
    if (conv) {
        constexpr int gen_mode = 1;  // A/B knob: 0 = stride-1 only
        general = d->stride != 1 || d->Hup != 0;
        if ((d->stride != 1 && d->stride != 2) || d->src_batch_mod != 0 || d->conv_asym_pad || d->Cin % CBK != 0 || (general && !gen_mode)) return 1;
        if (!general && (d->Hout != d->Hin || d->Wout != d->Win)) return 1;
        a_bytes = (d->M / ((int64_t)d->Hout * d->Wout)) * d->Hin * d->Win * (int64_t)d->Cin * 2;
    } else if (d->a_mode == APAD_A_PLAIN) {
        if (d->lda % 8 != 0) return 1;
        if (d->a2 != nullptr && (d->k_split % CBK != 0 || d->lda2 % 8 != 0)) return 1;
        // the tiled kernel sums these in K groups (gemm.hip: chosen by (N, K) alone so that a row's result does not depend on the
        // batch it rides in); this kernel sums k-tiles in order, so it must not take them for SOME row counts only
        if (d->K >= 384 && d->N >= 640) return 1;
        const int64_t rows_a = d->a_row_mod > 0 ? d->a_row_mod : d->M;
        a_bytes = ((rows_a - 1) * d->lda + (d->a2 ? d->k_split : d->K)) * 2;
    } else {
        return 1;
    }
    const int64_t w_bytes = ((d->N - 1) * d->ldw + d->K) * 2;
    int64_t a2_bytes = 0;
    if (!conv && d->a2 != nullptr) a2_bytes = (((d->a2_row_mod > 0 ? d->a2_row_mod : d->M) - 1) * d->lda2 + (d->K - d->k_split)) * 2;
    if (a_bytes >= (1LL << 31) || w_bytes >= (1LL << 31) || a2_bytes >= (1LL << 31)) return 1;
    CgP p;
    p.a = (const uint8_t*)d->a; p.w = (const uint8_t*)d->w; p.out = (uint8_t*)d->out; p.bias = (const uint8_t*)d->bias;
    p.residual = (const uint8_t*)d->residual; p.rg = (const uint8_t*)d->rowgroup_bias; p.step_ptr = d->step_ptr;
    p.ldo = d->ldo; p.ldr = d->ldr; p.ld_rg = d->ld_rg;
    p.M = (int32_t)d->M; p.N = (int32_t)d->N; p.K = (int32_t)d->K; p.lda = (int32_t)d->lda; p.ldw = (int32_t)d->ldw;
    p.Hin = d->Hin; p.Win = d->Win; p.Cin = d->Cin; p.res_mod = d->residual_row_mod;
    p.Hout = d->Hout; p.Wout = d->Wout; p.stride = d->stride; p.Hup = d->Hup; p.Wup = d->Wup;
    p.m_tiles = (int32_t)((d->M + CBM - 1) / CBM); p.n_tiles = (int32_t)(d->N / (sq ? 256 : 128));
    if (small) {
        p.m_tiles = (int32_t)((d->M + SBM - 1) / SBM);
        p.n_tiles = (int32_t)(d->N / SBN);
    }
    p.a_bytes = (uint32_t)a_bytes; p.w_bytes = (uint32_t)w_bytes;
    p.a2 = conv ? nullptr : (const uint8_t*)d->a2; p.lda2 = (int32_t)d->lda2; p.ksplit = d->k_split; p.a_mod = conv ? 0 : d->a_row_mod;
    p.a2_mod = d->a2_row_mod; p.a2_bytes = (uint32_t)a2_bytes;
    if (small) {
        p.a2 = nullptr; p.a_mod = 0;
        if (general) return d->dtype == APAD_BF16 ? cs_launch<APAD_BF16, true>(p, s) : cs_launch<APAD_F16, true>(p, s);
        return d->dtype == APAD_BF16 ? cs_launch<APAD_BF16, false>(p, s) : cs_launch<APAD_F16, false>(p, s);
    }
#define CG_GO(DT_)                                                                                              \
    if (general) return sq ? cg_launch<DT_, 2, 256>(p, s) : cg_launch<DT_, 2, 128>(p, s);                       \
    return sq ? (conv ? cg_launch<DT_, 1, 256>(p, s) : cg_launch<DT_, 0, 256>(p, s))                           \
              : (conv ? cg_launch<DT_, 1, 128>(p, s) : cg_launch<DT_, 0, 128>(p, s));
    if (d->dtype == APAD_BF16) { CG_GO(APAD_BF16) }
    CG_GO(APAD_F16)
#undef CG_GO
}
