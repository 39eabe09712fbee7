/*
 * apadapter_hip.h -- C ABI of libapadapter_hip.so: the MI355X (gfx950) implementation of the AP-adapter
 * audio-conditioned diffusion hot path.  Plain pointers and sizes only; no torch / C++ types.
 *
 * Every entry point enqueues on the caller's stream (a hipStream_t passed as void*; NULL = default
 * stream), allocates nothing, performs no host synchronisation and is therefore hipGraph-capturable.
 * All return 0 on success and a negative code on a rejected argument or launch failure; the message is
 * available from apad_last_error() (thread-local).  Nothing aborts the process.
 *
 * What each entry point replaces in the reference (paths relative to fundwotsai2001/AP-adapter):
 *   apad_attention      F.scaled_dot_product_attention x2 + blend in IPAttnProcessor2_0.__call__
 *                       (APadapter/ap_adapter/attention_processor.py:429-431, :443-445, :454) and the single
 *                       SDPA of AttnProcessor2_0.__call__ (:274-276); also timm Block attention inside
 *                       forward_encoder_no_random_mask_no_average (audio_encoder/models_mae.py:566-568)
 *   apad_gemm           attn.to_q/to_k/to_v/to_out, to_k_ip/to_v_ip (attention_processor.py:387,:406-407,
 *                       :435-436,:457); diffusers GEGLU/FeedForward, proj_in/proj_out, ResnetBlock2D /
 *                       Downsample2D / Upsample2D 3x3 convolutions and time_emb_proj
 *                       (pipeline/modeling_audioldm2.py:1032-1068 call sites); PatchEmbed_org.proj
 *                       (audio_encoder/models_mae.py:32-34,:42); timm Block qkv/proj/fc1/fc2
 *   apad_layernorm      diffusers BasicTransformerBlock norm1/2/3; timm Block norm1/2, encoder norm
 *   apad_groupnorm      Transformer2DModel.norm, ResnetBlock2D.norm1/norm2 (+SiLU), conv_norm_out
 *                       (pipeline/modeling_audioldm2.py:865-867)
 *   apad_audiomae_pool  AudioMAEConditionCTPoolRand.pool (audio_encoder/AudioMAE.py:148-182)
 *   apad_timestep_embedding  diffusers Timesteps (pipeline/modeling_audioldm2.py:317, :761)
 *   apad_cfg_ddim_step / apad_step_advance  CFG combine + DDIMScheduler.step
 *                       (pipeline/pipeline_audioldm2.py:1020-1025)
 */
#ifndef APADAPTER_HIP_H
#define APADAPTER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APAD_ABI_VERSION 10

/* element types of activations / weights */
enum { APAD_BF16 = 0, APAD_F16 = 1, APAD_F32 = 2 };

/* apad_gemm_desc.a_mode: how row m / column k of the A operand is fetched */
enum {
    APAD_A_PLAIN = 0,   /* A[m][k] = a[m*lda + k]                                                    */
    APAD_A_CONV3X3 = 1, /* implicit GEMM over NHWC: m=(b,oy,ox), k=(ky,kx,c); zero padding 1; optional
                           nearest-neighbour upsample of the source to (Hup,Wup) before the taps         */
    APAD_A_PATCH16 = 2, /* AudioMAE patch embed: a = fp32 mel [B,Hin,Win]; m=(b,ty,fx), k=(py,px)       */
    /* 3 is reserved (internal fast form of CONV3X3) */
    APAD_A_CONV1D = 4   /* HiFi-GAN: implicit GEMM over channels-last [B][Hin = T_in][Cin]; m=(b,t), k=(tap,c):
                           Conv1d (dilation, zero padding) or ConvTranspose1d (stride = up-sampling rate), with an
                           optional leaky-ReLU applied to the gathered input (the vocoder's pre-activations)     */
};

/* apad_gemm_desc.epilogue (applied to acc + bias + rowgroup_bias, before + residual) */
enum {
    APAD_EPI_NONE = 0, APAD_EPI_SILU = 1, APAD_EPI_GELU = 2, APAD_EPI_GEGLU = 3, APAD_EPI_TANH = 4,
    /* fp32 precision mode only (the prompt encoders, which run once per prompt): */
    APAD_EPI_RELU = 5,       /* CLAP's text projection                                                 */
    APAD_EPI_GELU_TANH = 6,  /* "gelu_new": 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) -- GPT-2    */
    APAD_EPI_GEGLU_TANH = 7  /* value * gelu_new(gate), weight rows value|gate -- T5's gated-gelu FF   */
};

/* apad_gemm_desc.out_mode */
enum {
    APAD_OUT_ROWMAJOR = 0, /* out[m*ldo + n]                                                          */
    APAD_OUT_VT = 1,       /* per-head transposed values for apad_attention: m=(b,l), n=(h,dd) ->
                              out[((b*heads + h)*head_dim + dd)*Lpad + l]                               */
    APAD_OUT_QKV = 2       /* fused q|k|v projection (N = 3*heads*head_dim): columns [0,C) -> out (row-major,
                              ldo), [C,2C) -> out2 (row-major, ldo), [2C,3C) -> out3 as APAD_OUT_VT            */
};

typedef struct apad_gemm_desc {
    const void* a;             /* activations (dtype, or fp32 for APAD_A_PATCH16)                      */
    const void* w;             /* weights [N][ldw] row-major, K contiguous (nn.Linear layout; conv
                                  weights as [Cout][ky][kx][Cin])                                       */
    void* out;
    const void* bias;          /* [N] or NULL                                                          */
    const void* residual;      /* [M][ldr] added after the activation, or NULL                         */
    const void* rowgroup_bias; /* [groups][ld_rg] added before the activation (time-embedding
                                  projection per sample), or NULL                                       */
    const int32_t* step_ptr;   /* optional device int32: rowgroup_bias row offset += *step_ptr         */
    int64_t M, N, K;           /* GEGLU: N = number of OUTPUT columns, w has 2N rows (value|gate)      */
    int64_t lda, ldw, ldo, ldr, ld_rg;
    int64_t rows_per_group;    /* rowgroup index = m / rows_per_group (+ *step_ptr)                    */
    int32_t a_mode, epilogue, out_mode, dtype;
    /* APAD_A_CONV3X3 / APAD_A_PATCH16 geometry */
    int32_t Hin, Win, Cin;     /* source spatial size and channels (NHWC)                              */
    int32_t Hout, Wout;        /* output spatial size                                                  */
    int32_t stride;            /* 1 or 2                                                               */
    int32_t Hup, Wup;          /* 0 = no upsample; else nearest upsample of source to (Hup,Wup)        */
    int32_t src_batch_mod;     /* 0 = off; else source batch = b % src_batch_mod (CFG duplication)     */
    int32_t residual_row_mod;  /* 0 = off; else residual row = m % residual_row_mod (e.g. pos_embed)    */
    /* APAD_OUT_VT / APAD_OUT_QKV */
    int32_t heads, head_dim, L, Lpad;
    void* out2;                /* APAD_OUT_QKV: k output                                                */
    void* out3;                /* APAD_OUT_QKV: v^T output                                              */
    /* APAD_A_CONV1D (Hin = T_in, Hout = T_out, Cin; K = taps * Cin; w = [Cout][tap][Cin]):
         transposed = 0: x[b][t*1 + tap*dilation - pad][c]                              (nn.Conv1d, stride 1)
         transposed = 1: x[b][(t + pad - tap) / stride][c] where that division is exact  (nn.ConvTranspose1d)   */
    int32_t taps, dilation, pad, transposed;
    int32_t a_pre_act;         /* 1: leaky_relu(a, a_pre_slope) on the gathered input                  */
    float a_pre_slope;
    int32_t conv_asym_pad;     /* APAD_A_CONV3X3: 0 = one zero row / column on every side (nn.Conv2d padding 1);
                                  1 = none on the top / left, one at the bottom / right (diffusers
                                  Downsample2D(padding=0): F.pad(x, (0,1,0,1)) then a stride-2 conv, the VAE encoder) */
    int32_t reserved_conv;
    /* LayerNorm folded into the contraction (16-bit dtypes).  diffusers' BasicTransformerBlock runs norm -> Linear three times per
       block; where no fused row-panel kernel covers the width (C = 640), the LayerNorm costs no launch and no pass over x:
         producer  (the GEMM that writes x, e.g. to_out + residual):  rowstat_out[m][N/64][2] = per 64-column block (sum, sum of
                   squares) of the stored row;
         consumer  (the Linear behind the LayerNorm):  a = x RAW, w = weight * gamma (caller-prepared, storage type),
                   out = epilogue( rstd_m * (a_m . w^T - mean_m * ln_colsum[n]) + ln_bias[n] ),
                   ln_colsum[n] = sum_k w[n][k] (of the scaled, rounded weight), ln_bias[n] = sum_k beta[k] W[n][k] + bias[n] (fp32),
                   mean_m / rstd_m summed in a fixed order from rowstat_in[m][0..rowstat_in_tiles). */
    float* rowstat_out;
    const float* rowstat_in;
    const float* ln_colsum;
    const float* ln_bias;
    int32_t rowstat_in_tiles;
    float ln_eps;
    /* APAD_A_PLAIN with a channel-concatenated A operand that is never materialised (ABI 6): columns [0, k_split) of row m come from
       a[(m % a_row_mod) * lda + k], columns [k_split, K) from a2[(m % a2_row_mod) * lda2 + k - k_split]; k_split % 64 == 0;
       a_row_mod / a2_row_mod = 0: no modulo.  The UNet's up blocks: torch.cat([hidden, skip], 1) in front of a resnet
       (modeling_audioldm2.py:1488) feeds its 1x1 shortcut convolution; a skip of the CFG-shared prefix holds one row set for both
       halves of the batch (row modulo). */
    const void* a2;
    int64_t lda2;
    int32_t k_split;
    int32_t a_row_mod, a2_row_mod;
    int32_t reserved_a2;
    void* out4;                /* APAD_OUT_QKV: optional v ROW-MAJOR [M][C] (ldo) beside the v^T of out3 -- the training step keeps both forms;
                                  NULL: not written (ABI 6) */
    const void* w_halo;        /* APAD_A_CONV3X3, optional (ABI 9): the same weights in the form apad_conv_halo_pack writes.  When it is given and the
                                  layer is inside the halo kernel's envelope (stride 1, zero padding 1, image width a power of two 4 .. 16, Cin % 64 == 0,
                                  N % 128 == 0, plain epilogue; a nearest-up-sampled source included), the convolution keeps the input pixels of a
                                  workgroup's image rows resident in LDS and fetches them ONCE per 64-channel chunk instead of once per filter tap
                                  (csrc/hconv.hip).  The choice depends on the layer only, never on the row count M.  Summation order: 64-channel
                                  chunk, tap, channel (w: tap, channel).  NULL: the im2col forms. */
    void* workspace;           /* w_halo layers that are summed in K slices (apad_conv_halo_workspace_bytes(M, N, Cin, Wout) > 0: the 2-pixel-wide
                                  images of the 64-token level): that many bytes of scratch for the fp32 slabs; NULL / too small: the im2col forms */
    int64_t workspace_bytes;
} apad_gemm_desc;

typedef struct apad_attn_desc {
    const void* q;         /* [B][N][H*D] via strides                                                  */
    const void* k;         /* segment 1 keys   [Bk][L][H*D] via strides                                */
    const void* vt;        /* segment 1 values, transposed [Bk][H][D][Lpad], zero padded               */
    const void* k2;        /* segment 2 (audio) keys or NULL                                           */
    const void* vt2;       /* segment 2 values                                                        */
    void* out;             /* [B][N][H*D] via strides                                                  */
    const float* key_bias; /* additive fp32 bias on segment-1 scores [B][L] (mask -> bias), or NULL    */
    float* lse;            /* optional out, single segment only: log2 sum_k exp2(log2e*(scale*s + bias)),
                              fp32 [B][H][round_up(N,32)] -- what apad_attention_bwd recomputes P from   */
    int64_t q_stride_b, q_stride_n;
    int64_t k_stride_b, k_stride_l, vt_stride_b;
    int64_t k2_stride_b, k2_stride_l, vt2_stride_b;
    int64_t o_stride_b, o_stride_n;
    int32_t B, N, H, D;
    int32_t L, Lpad, L2, Lpad2; /* L2 = 0: single softmax segment                                      */
    int32_t kv_batch_div;       /* K/V batch index = b / kv_batch_div (1 = one K/V set per sample)     */
    int32_t kv2_batch_div;
    int32_t dtype;
    float softmax_scale;        /* 1/sqrt(D)                                                           */
    float scale2;               /* out = softmax1.V1 + scale2 * softmax2.V2  (ap_scale)                */
    int32_t q_prescaled;        /* 1: q already carries softmax_scale * log2(e) (e.g. the to_q weight rows were scaled
                                   before the projection, so q is rounded to the storage type ONCE): scores are used as
                                   base-2 exponents directly and softmax_scale is ignored                    */
} apad_attn_desc;

/* Row-panel GEMM: the fused projection kernel of the transformer blocks.  Each workgroup keeps a 128-row panel of
 * x (all K <= 640 columns) in registers -- optionally LayerNorm-ed in place (BasicTransformerBlock norm1/2/3) --
 * and streams the weight rows through LDS, so x is read from HBM once for ALL output columns (q|k|v fused, or the
 * 8C-wide GEGLU projection) and LayerNorm costs no extra pass.  Up to 3 column segments with their own output
 * (row-major, or the per-head transposed V^T apad_attention consumes).
 * Replaces: norm + attn.to_q/to_k/to_v (attention_processor.py:387,:406-407), to_out[0] (:457), diffusers
 * GEGLU.proj, Transformer2DModel.proj_in/proj_out. */
typedef struct apad_rp_segment {
    void* out;
    const void* bias;   /* [n_cols] or NULL (GEGLU: [2*n_cols], value|gate)                                */
    int64_t ldo;        /* row stride of a row-major output                                                */
    int32_t n_cols;     /* output columns of this segment (multiple of 64)                                 */
    int32_t mode;       /* APAD_OUT_ROWMAJOR or APAD_OUT_VT                                                */
} apad_rp_segment;

typedef struct apad_rp_desc {
    const void* x;        /* [M][lda]                                                                      */
    const void* w;        /* [sum n_cols (x2 for GEGLU)][ldw], rows in segment order                       */
    const void* ln_gamma; /* [K] or NULL: LayerNorm(x) before the projection                               */
    const void* ln_beta;
    const void* residual; /* [M][ldr] or NULL; single row-major segment only                               */
    int64_t M, lda, ldw, ldr;
    int32_t K;            /* 256 or 384 (else -3)                                                               */
    int32_t epilogue;     /* APAD_EPI_NONE / SILU / GELU / GEGLU                                           */
    int32_t dtype, n_segments;
    float ln_eps;
    int32_t heads, head_dim, L, Lpad; /* APAD_OUT_VT segments: m = (b, l)                                  */
    apad_rp_segment seg[3];
} apad_rp_desc;

/* Fused feed-forward of a BasicTransformerBlock (diffusers FeedForward with GEGLU, dropout 0):
 *     out = x + W2 . (value * gelu(gate)) + b2,   [value | gate] = W1 . LayerNorm(x) + b1
 * in one kernel; the 8C projection and the 4C activation never leave registers.  C in {256, 384} (else -3). */
typedef struct apad_mlp_desc {
    const void* x;        /* [M][C]: un-normalised hidden states, also the residual                         */
    const void* ln_gamma; /* [C] or NULL (no LayerNorm)                                                     */
    const void* ln_beta;
    const void* w1;       /* [8C][C]: GEGLU.proj weight, value rows then gate rows                          */
    const void* b1;       /* [8C] or NULL                                                                   */
    const void* w2;       /* [C][4C]: FeedForward.net[2] weight                                             */
    const void* b2;       /* [C] or NULL                                                                    */
    void* out;            /* [M][C]; may alias x                                                            */
    int64_t M;
    int32_t C, dtype;
    float ln_eps;
    int32_t reserved;
} apad_mlp_desc;

/* Fused cross-attention sub-layer for hoisted, short key / value sets (<= 64 keys per segment):
 *     out = x + to_out( A(q, K1, V1, key_bias) [+ scale2 * A(q, K2, V2)] ) + bo,   q = to_q(LayerNorm(x))
 * = IPAttnProcessor2_0.__call__ (attention_processor.py:347-470) / AttnProcessor2_0.__call__ (:214-294) together with
 * the block's pre-LayerNorm and residual, in ONE launch.  Envelope: C = 256, 8 heads (else -3).
 * The kernel is weight-stationary: every wave keeps its 32 rows of to_q / to_out[0] in registers, so the two weights
 * and the hoisted K / V are passed FRAGMENT-PACKED (each MFMA operand fragment = one contiguous KB):
 *   apad_xattn_pack_weight  once per weight (re-pack when the parameter changes)
 *   apad_xattn_pack_kv      once per hoisted K / V^T set (i.e. per pipeline call, with the K/V projection itself) */
typedef struct apad_xattn_desc {
    const void* x;          /* [B*N][C] un-normalised hidden states (also the residual)                     */
    const void* ln_gamma;   /* [C] or NULL: LayerNorm(x) in front of to_q -- applied BY ALGEBRA (ABI 8): see q_fold */
    const void* ln_beta;
    const void* wq_packed;  /* apad_xattn_pack_weight(attn.to_q.weight); with a LayerNorm: of W' = round(W * gamma)   */
    const void* wo_packed;  /* apad_xattn_pack_weight(attn.to_out[0].weight)                                */
    const void* bo;         /* [C] attn.to_out[0].bias or NULL                                              */
    const void* kv1_packed; /* apad_xattn_pack_kv(to_k(tokens), to_v(tokens)^T) of segment 1 (text / T5)     */
    const float* key_bias;  /* [B][L1] fp32 additive bias on segment 1, or NULL                             */
    const void* kv2_packed; /* segment 2 (to_k_ip / to_v_ip of the audio tokens) or NULL                     */
    void* out;              /* [B*N][C]                                                                     */
    /* ABI 8: the tile keeps the RAW rows of x in LDS from the first phase to the last (the residual is added in place, x is read from HBM
     * once), so the LayerNorm is folded into the q-projection: q = rstd * (W' x - mean * q_fold[0]) + q_fold[1] with W' = round(W * gamma)
     * passed as wq_packed, q_fold[0][c] = sum_k W'[c][k] (fp32 sums of the ROUNDED weights), q_fold[1][c] = sum_k W[c][k] * beta[k].
     * Required when ln_gamma != NULL (fp32 [2][C]); NULL when there is no LayerNorm. */
    const float* q_fold;
    int32_t B, N, C, heads;
    int32_t L1, L2;         /* L2 = 0: single segment                                                       */
    int32_t dtype, reserved;
    float ln_eps, softmax_scale, scale2, reserved_f;
} apad_xattn_desc;
int apad_sizeof_xattn_desc(void);
int apad_echo_xattn_desc(const apad_xattn_desc* d, double* out, int cap);
int apad_fused_cross_attention(const apad_xattn_desc* d, void* stream);

/* The same sub-layer for the 384-wide level (252 tokens per sample, 8 heads of 48), where the weights do not fit the weight-stationary
 * registers of apad_fused_cross_attention: 64-token row tiles stay in LDS through LayerNorm -> to_q -> attention -> to_out -> + residual
 * (attention.hip, xattn_rows_kernel).  K / V sets as apad_attention takes them (k [B][L][C] row-major, vt [B][heads][d][Lpad] zero-padded);
 * <= 64 keys per segment, or <= 512 in segment 2 beside <= 32 in segment 1 (8 text + 128 audio keys: the timbre / accompaniment presets; 129 .. 512: pooling 1
 * and the mixed poolings, run in 64-key chunks with a running maximum / sum, ABI 10); longer segments: the un-fused chain.
 * ABI 10: a segment may instead be given FRAGMENT-PACKED (apad_rows_pack_kv below): kN = the packed set, vtN = NULL, LpadN unused -- every fragment
 * load of the attention phase is then one contiguous KB instead of 32 - 64 cache lines (the K / V sets are timestep-invariant: packed once per site).
 * The two weights FRAGMENT-PACKED:
 *   packed[(rt * (C / 16) + ks) * 512 + lane * 8 + e] = W[rt * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + e]   (elements)
 * i.e. W.view(C/32, 32, C/16, 2, 8).permute(0, 2, 3, 1, 4): every MFMA operand fragment is one contiguous KB.  Envelope: C = 384 (64-token
 * tiles) or 640 (32-token tiles; the Python side routes it only on request), 8 heads, 16-bit (else -3).  Replaces, per site, to_q + scaled_dot_product_attention (x 2 for the adapter) + to_out[0] of attention_processor.py:387-457
 * / :256-289 plus the block's norm2 / norm3 and residual add (modeling_audioldm2 BasicTransformerBlock).  (ABI 6) */
typedef struct apad_xrows_desc {
    const void* x;         /* [B*N][C] un-normalised hidden states (also the residual)  */
    const void* ln_gamma;  /* [C] or NULL                                               */
    const void* ln_beta;
    const void* wq_packed; /* attn.to_q.weight, fragment-packed                         */
    const void* wo_packed; /* attn.to_out[0].weight, fragment-packed                    */
    const void* bo;        /* [C] or NULL                                               */
    const void* k1;        /* [B][L1][C]; or the apad_rows_pack_kv set with vt1 = NULL  */
    const void* vt1;       /* [B][heads][C / heads][Lpad1]                              */
    const float* key_bias; /* [B][L1] fp32 additive bias on segment 1, or NULL          */
    const void* k2;        /* segment 2 (to_k_ip / to_v_ip of the audio tokens) or NULL */
    const void* vt2;
    void* out;             /* [B*N][C]; may alias x                                     */
    int32_t B, N, C, heads;
    int32_t L1, Lpad1, L2, Lpad2;
    int32_t dtype, reserved;
    float ln_eps, softmax_scale, scale2, reserved_f;
} apad_xrows_desc;
int apad_sizeof_xrows_desc(void);
int apad_cross_attention_rows(const apad_xrows_desc* d, void* stream);
/* (ABI 10) One key / value set of a cross-attention site -- k [B][L][heads * head_dim] row-major, vt [B][heads][head_dim][Lpad], i.e. what to_k / to_v
 * (to_k_ip / to_v_ip) of attention_processor.py:256-259 / :432-433 produce, in apad_attention's layout -- re-ordered into the MFMA operand fragments
 * apad_cross_attention_rows and apad_hs_attention read, per (sample, head), NU = ceil(L / 32):
 *   K   fragment (u < NU, cc < head_dim / 16):          lane -> k[key = min(32 u + lane % 32, L - 1)][16 cc + 8 (lane / 32) .. + 8]
 *   V^T fragment (st < 2 NU, dt < ceil(head_dim / 32)): lane -> vt[d = 32 dt + lane % 32][16 st + 4 (lane / 32) + {0..3, 8..11}]   (zeros for d >= head_dim)
 * 1 KB each, K fragments first; apad_rows_packed_kv_bytes = B * heads * NU * (head_dim / 16 + 2 ceil(head_dim / 32)) KB.  head_dim % 16 == 0, 16-bit. */
int64_t apad_rows_packed_kv_bytes(int32_t B, int32_t heads, int32_t head_dim, int32_t L);
int apad_rows_pack_kv(const void* k, const void* vt, void* out, int32_t B, int32_t heads, int32_t head_dim, int32_t L, int32_t Lpad, int32_t dtype, void* stream);
/* The attention sub-layers of the 64-token level (C = 640, 8 heads of 80, <= 64 tokens per sample) in TWO launches (hsattn.hip; ABI 7):
 *   apad_hs_attention  workgroup = (sample, head pair): LayerNorm(x) (normalisation here, affine part folded into the weights) -> the pair's q|k|v (self-attention) or q (cross-attention over the
 *                      hoisted K / V^T sets of apad_attention's layout) -> softmax attention of its two heads (one segment, a masked
 *                      segment, or the adapter's text + scale2 * audio pair) -> O[:, pair's 160 columns]
 *   apad_hs_out        workgroup = (sample, output-column quarter): out = residual + (O . Wo^T + bias), plus the per-32-column (sum, sum of
 *                      squares) row statistics [B*N][20][2] a folded LayerNorm (apad_gemm_desc.rowstat_in, 20 tiles) reads
 * Every workgroup streams a disjoint quarter of the weights once against its sample's tokens in LDS.  Weights FRAGMENT-PACKED per
 * quarter (one contiguous KB per MFMA operand fragment):
 *   self-attention  w_packed[((p * 15 + t) * 40 + ks) * 512 + lane * 8 + e] = Wqkv[(t / 5) * 640 + p * 160 + (t % 5) * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + e]
 *                   with Wqkv = [to_q ; to_k ; to_v] stacked (1920 x 640; q_prescaled = 1: the to_q rows carry log2(e) / sqrt(80))
 *   cross / to_out  w_packed[((p * 5 + t) * 40 + ks) * 512 + lane * 8 + e]  = W[p * 160 + t * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + e]
 * Replaces, per site, norm1 / norm2 + to_q / to_k / to_v + scaled_dot_product_attention (x 2 for the adapter) (attention_processor.py:256-276,
 * :387-454) and to_out[0] + the block's residual add (:279, :457; modeling_audioldm2 BasicTransformerBlock).  Outside the envelope: -3. */
typedef struct apad_hs_attn_desc {
    const void* x;         /* [B*N][640] hidden states                                                */
    const void* w_packed;  /* see above                                                               */
    const float* w_bias;   /* [4][NTILE * 32] fp32 in the packed row order (W . ln_beta + bias), or NULL */
    const void* k1;        /* cross: [B][L1][640]; self: NULL.  ABI 10: or the segment's apad_rows_pack_kv set with vt1 = NULL (also k2 / vt2) */
    const void* vt1;       /* cross: [B][8][80][Lpad1]                                                */
    const float* key_bias; /* cross: [B][L1] fp32 additive bias on segment 1, or NULL                 */
    const void* k2;        /* cross: segment 2 (to_k_ip / to_v_ip of the audio tokens) or NULL        */
    const void* vt2;
    void* out;             /* O [B*N][640]                                                            */
    int32_t B, N, C, heads;
    int32_t L1, Lpad1, L2, Lpad2;
    int32_t self_attention, q_prescaled, dtype;
    int32_t normalize;     /* 1: rows are normalised ((x - mean) * rstd, eps = ln_eps) before the projection; the LayerNorm's gamma is
                              folded into w_packed and W . beta into w_bias by the caller (apad_gemm's folded-LayerNorm algebra)  */
    float ln_eps, softmax_scale, scale2, reserved_f;
} apad_hs_attn_desc;
int apad_sizeof_hs_attn_desc(void);
int apad_hs_attention(const apad_hs_attn_desc* d, void* stream);
typedef struct apad_hs_out_desc {
    const void* o;         /* [B*N][640] attention output (all heads)                                 */
    const void* w_packed;  /* to_out[0].weight, packed as above                                       */
    const void* bias;      /* [640] or NULL                                                           */
    const void* residual;  /* [B*N][640] or NULL                                                      */
    void* out;             /* [B*N][640]; may alias residual                                          */
    float* rowstat_out;    /* [B*N][20][2] fp32 or NULL                                               */
    int32_t B, N, C, dtype;
} apad_hs_out_desc;
int apad_sizeof_hs_out_desc(void);
int apad_hs_out(const apad_hs_out_desc* d, void* stream);
/* the feed-forward's second Linear at that level with the same descriptor: o = H [B*N][2560], w_packed = W2 [640][2560] packed per output quarter
 *   w_packed[((q * 5 + t) * 160 + ks) * 512 + lane * 8 + e] = W2[q * 160 + t * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + e]
 * out = residual + (H . W2^T + bias); the four K-quarters of a workgroup are summed in a fixed order */
int apad_hs_ff2(const apad_hs_out_desc* d, void* stream);
/* LayerNorm + to_q | to_k | to_v + softmax self-attention in ONE launch, workgroup = (sample, head): K / V^T of the whole sample projected into LDS,
 * Q kept in registers, the key loop over the resident tiles (attention.hip, sattn_fused_kernel; ABI 7).  Envelope: C = 256 / 8 heads / N <= 1024 and
 * C = 384 / 8 heads / N <= 256, 16-bit (else -3).  The LayerNorm is applied by algebra (gamma folded into the weights by the caller):
 *   the head's 3 d rows [to_q rows h d .. | to_k rows | to_v rows] of W' = [to_q * log2(e) / sqrt(d) ; to_k ; to_v] * gamma, zero-padded to T3 = ceil(3 d / 32)
 *   row tiles: virtual row v = j * 32 + (lane & 31)  ->  W'[(v / d) * C + h * d + v % d]
 *   w_packed[((h * T3 + j) * (C / 16) + ks) * 512 + lane * 8 + e] = that row's [ks * 16 + (lane >> 5) * 8 + e]
 *   colsum_bias[(h * 2 + 0) * T3 * 32 + v] = sum_k of the ROUNDED packed row (fp32);   colsum_bias[(h * 2 + 1) * T3 * 32 + v] = (W . beta) of that row
 * x [B*N][C] un-normalised -> out O [B*N][C] (all heads; to_out + residual: apad_rowpanel_gemm as before).
 * Replaces norm1 + to_q / to_k / to_v + scaled_dot_product_attention of attention_processor.py:256-276. */
int apad_self_attention_fused(const void* x, const void* w_packed, const float* colsum_bias, void* out, int32_t B, int32_t N, int32_t C, int32_t heads,
                              float ln_eps, int32_t dtype, void* stream);
/* The feed-forward's GEGLU projection at the same level (diffusers GEGLU behind norm3: H = value * gelu(gate), [value | gate] = Linear(640 -> 5120)),
 * workgroup = (sample, hidden quarter), rows normalised in the launch (normalize = 1; gamma / beta folded into w_packed / w_bias like above):
 *   w_packed[(((q * 20 + t) * 2 + j) * 40 + ks) * 512 + lane * 8 + e] = W[j * 2560 + q * 640 + t * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + e]
 *   w_bias[((q * 20 + t) * 2 + j) * 32 + r] (fp32) = (W . beta + b)[j * 2560 + q * 640 + t * 32 + r]          x [B*N][640] -> out H [B*N][2560]      (ABI 7) */
int apad_hs_geglu(const void* x, const void* w_packed, const float* w_bias, void* out, int32_t B, int32_t N, int32_t C, int32_t normalize, float ln_eps,
                  int32_t dtype, void* stream);
/* The same feed-forward from PACKED weights (ABI 8; csrc/mlp3.hip): a wave owns 64 tokens and all C output columns, so every weight fragment read from LDS
 * feeds two MFMAs; the weights travel L2 -> LDS by DMA in exactly the order the loop consumes them.  apad_mlp_pack builds that stream once per
 * FeedForward (re-pack when a parameter changes):
 *   w_packed: 66 stages of 24 KB; stage i = 16 fragments of GEGLU.proj rows (value units 16 i .. 16 i + 15, then their gate rows) x k-steps of 16, then
 *             8 fragments of net[2] columns of units 16 (i - 2) .. in the order (j & 3) + 8 (j >> 2) + 4 half (the MFMA C layout of the first GEMM
 *             read as the B operand of the second); one fragment = 64 lanes x 8 elements = 1 KB; stages past either end are zero
 *   b1_packed: fp32 [66][2][16], b1 in the C-layout register order of the lane half
 * d->w1 / d->b1 / d->w2 are ignored; d->b2, the LayerNorm and the residual (= x) as in apad_geglu_mlp.  C = 256 (else -3). */
int64_t apad_mlp_packed_bytes(int32_t C);
int64_t apad_mlp_packed_bias_floats(int32_t C);
int apad_mlp_pack(const void* w1, const void* b1, const void* w2, void* w_packed, float* b1_packed, int32_t C, int32_t dtype, void* stream);
int apad_geglu_mlp_packed(const apad_mlp_desc* d, const void* w_packed, const float* b1_packed, void* stream);
/* LayerNorm + the GEGLU projection of a feed-forward at C = 384 from PACKED weights (ABI 8; csrc/geglu3.hip): H [M][4C] = value * gelu(gate),
 * [value | gate] = W1 . LayerNorm(x) + b1 (diffusers GEGLU behind norm3; ln_gamma / ln_beta NULL: no LayerNorm) on the 64-token register-block loop
 * of apad_geglu_mlp_packed without its second GEMM (4C -> C stays an apad_gemm).  apad_geglu_pack builds the stream once per FeedForward:
 *   w_packed: [4 hidden quarters][24 chunks of 16 units][24 k-steps][64 lanes][8]: fragment rows 0..15 the chunk's value units, 16..31 their gate rows
 *   b1_packed: fp32 [4][24][2][16], b1 in the C-layout register order of the lane half
 * Bit-equal to apad_rowpanel_gemm's GEGLU epilogue (the form smaller launches take).  C = 384 (else -3). */
int64_t apad_geglu_packed_bytes(int32_t C);
int64_t apad_geglu_packed_bias_floats(int32_t C);
int apad_geglu_pack(const void* w1, const void* b1, void* w_packed, float* b1_packed, int32_t C, int32_t dtype, void* stream);
int apad_layernorm_geglu_packed(const void* x, const void* ln_gamma, const void* ln_beta, const void* w_packed, const float* b1_packed, void* out, int64_t M,
                                int32_t C, float ln_eps, int32_t dtype, void* stream);
/* w [256][ldw] (nn.Linear layout) -> packed [8 row slices][16 k-steps][64 lanes][8], 128 KB */
int apad_xattn_pack_weight(const void* w, void* packed, int64_t ldw, int32_t dtype, void* stream);
/* bytes of the packed form of one segment's K / V^T: B * 8 heads * ceil(L/32) * 4 KB */
int64_t apad_xattn_packed_kv_bytes(int32_t B, int32_t L);
/* k [B][L][256] (strides in elements), vt [B][8][32][Lpad] (apad_gemm APAD_OUT_VT) -> packed; L <= 64 */
int apad_xattn_pack_kv(const void* k, const void* vt, void* packed, int32_t B, int32_t L, int32_t Lpad, int64_t k_stride_b,
                       int64_t k_stride_l, int64_t vt_stride_b, int32_t dtype, void* stream);

const char* apad_last_error(void);
int apad_abi_version(void);
/* process-wide choice of apad_gemm's kernel for latency-bound plain launches (64 x 64 tiles): 0 = tiled, 1 / 2 = LDS-DMA ring form for
 * under-filled grids / for every launch below 16000 rows, -1 = the APAD_GEMM_RING environment variable (default 0).  Same results bit for
 * bit; the training step switches it on (one stream, ~2600 such launches per step).  Returns the previous setting.  (ABI 6) */
int apad_set_gemm_ring(int32_t mode);
/* size of the descriptor structs as compiled, for binding self-checks */
int apad_sizeof_gemm_desc(void);
int apad_sizeof_attn_desc(void);
/* writes every field of the descriptor as a double into out[]; returns the number written (binding check,
   host only, no GPU work) */
int apad_echo_gemm_desc(const apad_gemm_desc* d, double* out, int cap);
int apad_echo_attn_desc(const apad_attn_desc* d, double* out, int cap);
int apad_sizeof_rp_desc(void);
int apad_echo_rp_desc(const apad_rp_desc* d, double* out, int cap);
int apad_sizeof_mlp_desc(void);
int apad_echo_mlp_desc(const apad_mlp_desc* d, double* out, int cap);

int apad_gemm(const apad_gemm_desc* d, void* stream);
/* apad_gemm_desc::w_halo: w [N][ky][kx][Cin] (the conv form of apad_gemm_desc::w) -> out, the same N * 9 * Cin elements as
   [Cin / 64][tap][32-channel half][N][32 channels] with the four 16-byte slots of every 64-byte record XOR-ed by (n >> 2) & 3 (the LDS image the
   kernel's weight stages are copied into verbatim).  Cin % 64 == 0; 16-bit dtypes.  One-off, per parameter version.
   Replaces nothing in the reference: it is a re-layout of ResnetBlock2D.conv1 / conv2 / Upsample2D.conv weights (modeling_audioldm2.py call sites
   as for apad_gemm). */
int apad_conv_halo_pack(const void* w, void* out, int64_t N, int64_t Cin, int32_t dtype, void* stream);
/* bytes apad_conv_halo_pack writes: N * 9 * Cin elements in the wide form; N <= 16 (the narrow form of csrc/hconv.hip: conv_out, stationary weights as
   16-row MFMA A-operand fragments, rows >= N zero): Cin / 64 * 18 KB */
int64_t apad_conv_halo_packed_bytes(int64_t N, int64_t Cin);
/* scratch a w_halo convolution of this geometry needs in apad_gemm_desc::workspace (0: none) */
int64_t apad_conv_halo_workspace_bytes(int64_t M, int64_t N, int64_t Cin, int32_t Wout);
/* measurement probe (no reference counterpart): a memory-free stream of dense bf16 MFMAs on zero (mode 0) or pseudo-random (mode 1) operands on 512
   workgroups; *flops = the FLOPs of the launch.  bench.py times it to state the matrix-pipe rate THIS device delivers on real operand data next to the
   nominal peak (the rate is power-managed and depends on the data: tools/ubench/mfma_data.hip). */
int apad_probe_mfma(void* sink, int32_t mode, int32_t iters, double* flops, void* stream);
/* diagnostic: how many apad_gemm calls of this process went to the halo kernel (tests assert the route with it) */
int64_t apad_hconv_launch_count(void);
int apad_attention(const apad_attn_desc* d, void* stream);
/* returns -3 (and sets the error text) when the shape is outside the kernel's envelope; callers then use apad_gemm */
int apad_rowpanel_gemm(const apad_rp_desc* d, void* stream);
/* returns -3 when C is outside {256, 384}; callers then use apad_rowpanel_gemm / apad_gemm for the two halves */
int apad_geglu_mlp(const apad_mlp_desc* d, void* stream);

int apad_layernorm(const void* x, const void* gamma, const void* beta, void* out, int64_t M, int32_t C,
                   int64_t ldx, int64_t ldo, float eps, int32_t dtype, void* stream);

/* GroupNorm over NHWC x[B][HW][C] with G groups, optional fused SiLU.  workspace: fp32, at least
   apad_groupnorm_workspace_bytes(B, HW, G) bytes. */
int64_t apad_groupnorm_workspace_bytes(int32_t B, int32_t HW, int32_t G);
int apad_groupnorm(const void* x, const void* gamma, const void* beta, void* out, void* workspace, int32_t B,
                   int32_t HW, int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype, void* stream);
/* GroupNorm (+ SiLU) over the channel concatenation [xa | xb] without materialising it (ABI 6): xa [Ba][HW][Ca], xb [Bb][HW][Cb],
 * out [B][HW][Ca + Cb]; sample b reads rows of xa / xb at batch index b % Ba / b % Bb (a skip tensor of the CFG-shared prefix is
 * stored once for both halves of the batch).  Ca % 8 == 0, Cb % 8 == 0; xb = NULL / Cb = 0: apad_groupnorm.  16-bit dtypes. */
int apad_groupnorm2(const void* xa, const void* xb, const void* gamma, const void* beta, void* out, void* workspace, int32_t B,
                    int32_t Ba, int32_t Bb, int32_t HW, int32_t Ca, int32_t Cb, int32_t G, float eps, int32_t silu, int32_t dtype,
                    void* stream);

/* rep [B][513][768] (dtype) -> out [B][(64/tp)*(8/fp)][768] (out_dtype): drop CLS, (avg + max)/2 */
int apad_audiomae_pool(const void* rep, void* out, int32_t B, int32_t tp, int32_t fp, int32_t dtype,
                       int32_t out_dtype, void* stream);

/* t [n] fp32 -> out [n][dim] (dtype): sinusoidal embedding, diffusers Timesteps semantics */
int apad_timestep_embedding(const float* t, void* out, int32_t n, int32_t dim, int32_t flip_sin_to_cos,
                            float freq_shift, int32_t dtype, void* stream);

/* eps2 [2B][n] (dtype, unconditional half first); latents [B][n] fp32 updated in place;
   unet_in [B][n] (dtype) receives the new latents; eps_out (optional, fp32 [B][n]) the guided noise.
   coef [steps][2] fp32: x_prev = coef[s][0]*x + coef[s][1]*eps, s = *step_ptr. */
int apad_cfg_ddim_step(const void* eps2, float* latents, void* unet_in, float* eps_out, const float* coef,
                       const int32_t* step_ptr, float guidance_scale, int32_t B, int64_t n, int32_t dtype,
                       void* stream);
int apad_step_advance(int32_t* step_ptr, void* stream);
/* out = (a + b + c) * scale, element-wise over n values of `dtype` (HiFi-GAN: mean of the three residual-block branches,
   SpeechT5HifiGan.forward) */
int apad_mix3(const void* a, const void* b, const void* c, void* out, int64_t n, float scale, int32_t dtype, void* stream);

/* Row softmax: out[m][n] = softmax over n of (scale * x[m][n] + bias[m][n]), fp32 statistics, one wave per row (x, out: [M][ld]
   of dtype; bias: optional fp32 [M][ldb] -- T5's relative-position bias, causal / padding masks as -inf).
   The VAE mid-block attention (diffusers AutoencoderKL decoder/encoder, one head of d = 512 over the 4000 latent pixels --
   outside apad_attention's head-dim envelope) runs as apad_gemm (Q.K^T) -> apad_softmax_rows -> apad_gemm (P.V). */
int apad_softmax_rows(const void* x, const float* bias, void* out, int64_t M, int32_t N, int64_t ldx, int64_t ldb, int64_t ldo,
                      float scale, int32_t dtype, void* stream);

/* T5LayerNorm / RMS norm (mode 0): out = x * rsqrt(mean(x^2) + eps) * gamma; F.normalize (mode 1, gamma ignored):
   out = x / max(||x||_2, eps).  x, out [M][ld] of dtype, one wave per row, fp32 statistics. */
int apad_rmsnorm(const void* x, const void* gamma, void* out, int64_t M, int32_t C, int64_t ldx, int64_t ldo, float eps,
                 int32_t mode, int32_t dtype, void* stream);

/* nn.Embedding lookup: out[i][:] = table[ids[i]][:] (ids int64 on the device, rows of C elements of dtype); an id outside
   [0, rows) writes zeros. */
int apad_gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int64_t rows, int32_t C, int32_t dtype,
                     void* stream);

/* The VAE encoder's posterior draw (diffusers DiagonalGaussianDistribution.sample(), train_apadapter_v2.py:895-897):
   moments [rows][2*latent] = (mean | logvar) per latent pixel, noise [rows][latent] ->
   out [rows][latent] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale. */
int apad_gaussian_sample(const void* moments, const void* noise, void* out, int64_t rows, int32_t latent, float scale,
                         int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Training step of the adapter (SURVEY a-11; train_apadapter_v2.py:941-979).  The UNet is frozen: only INPUT
 * gradients flow through its layers, and the only weight gradients are those of to_k_ip / to_v_ip
 * (attention_processor.py:324-325).  GEMM-shaped gradients reuse apad_gemm:
 *   linear dgrad     dx = dy . W            -> apad_gemm(a = dy, w = W^T [K][N] contiguous)
 *   conv3x3 dgrad    stride 1: apad_gemm conv mode on dy with w'[ci][(2-ky,2-kx,co)] = w[co][(ky,kx,ci)];
 *                    stride 2: apad_zero_stuff2 first; nearest-upsampled source: apad_upsample_nearest_bwd after
 *   adapter wgrad    dW[C][768] = dK^T . ehs -> apad_transpose_pad both operands, then apad_gemm
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct apad_attn_bwd_desc {
    const void* q;      /* [B][N][H*D] contiguous                                                       */
    const void* k;      /* [B][L][H*D]                                                                  */
    const void* v;      /* [B][L][H*D] row-major values                                                 */
    const void* qt;     /* Q^T  [B][H][D][Npad] (apad_head_transpose); only read when dk/dv are wanted   */
    const void* kt;     /* K^T  [B][H][D][Lpad]                                                         */
    const void* out;    /* forward output of THIS segment [B][N][H*D]                                   */
    const void* dout;   /* gradient of the blended output [B][N][H*D]                                   */
    const void* doutt;  /* dout^T [B][H][D][Npad]; only for dk/dv                                       */
    const float* lse;   /* [B][H][Npad] from apad_attention                                             */
    const float* key_bias; /* [B][L] fp32 additive bias of the forward, or NULL                         */
    float* delta;       /* workspace fp32 [B][H][Npad]                                                  */
    void* dq;           /* [B][N][H*D]                                                                  */
    void* dk;           /* [B][L][H*D] or NULL (keys / values that come from frozen conditioning)       */
    void* dv;
    int32_t B, N, H, D, L, Npad, Lpad, dtype;
    float softmax_scale;
    float dout_scale;      /* the segment saw dout_scale * dout (ap_scale for the audio branch, :454)    */
    int32_t accumulate_dq; /* 1: dq += (second segment of the decoupled cross-attention)                */
    int32_t ld_grad;       /* row stride (elements) of dq / dk / dv; 0 = H*D.  3*H*D: the three are column blocks of ONE [B*N][3C]
                              buffer (self-attention: the input-gradient GEMM of q|k|v reads it without a concatenation).  16-bit only */
} apad_attn_bwd_desc;
int apad_sizeof_attn_bwd_desc(void);
int apad_echo_attn_bwd_desc(const apad_attn_bwd_desc* d, double* out, int cap);
int apad_attention_bwd(const apad_attn_bwd_desc* d, void* stream);
/* x [B][N][H*D] -> xt [B][H][D][pad], zero-filled for n >= N (pad % 32 == 0) */
int apad_head_transpose(const void* x, void* xt, int32_t B, int32_t N, int32_t H, int32_t D, int32_t pad, int32_t dtype,
                        void* stream);
/* the same for up to three tensors of ONE shape in one launch (q, k and dO of a self-attention backward); unused pairs NULL */
int apad_head_transpose3(const void* x0, void* xt0, const void* x1, void* xt1, const void* x2, void* xt2, int32_t B, int32_t N,
                         int32_t H, int32_t D, int32_t pad, int32_t dtype, void* stream);

int apad_layernorm_bwd(const void* x, const void* gamma, const void* dy, void* dx, int64_t M, int32_t C, float eps,
                       int32_t dtype, void* stream);
/* the same with the gradient that reaches x past the LayerNorm added in (dres, may be NULL): the pre-norm sub-layers' residual
 * connection (x feeds LayerNorm AND the residual add), whose two gradients the autograd engine would otherwise sum with a kernel of
 * its own; dx = round(ln_bwd) + dres (ABI 6) */
int apad_layernorm_bwd_add(const void* x, const void* gamma, const void* dy, const void* dres, void* dx, int64_t M, int32_t C, float eps,
                           int32_t dtype, void* stream);
int apad_groupnorm_bwd(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, int32_t B, int32_t HW,
                       int32_t C, int32_t G, float eps, int32_t silu, int32_t dtype, void* stream);
/* proj [M][2N] (value | gate) -> h [M][N] = value * gelu(gate), and its gradient dproj [M][2N] */
int apad_geglu(const void* proj, void* h, int64_t M, int32_t N, int32_t dtype, void* stream);
int apad_geglu_bwd(const void* proj, const void* dh, void* dproj, int64_t M, int32_t N, int32_t dtype, void* stream);
/* dup [B][Hup*Wup][C] -> dx [B][H*W][C]: adjoint of the nearest-neighbour gather floor(dst*in/out) of apad_gemm */
int apad_upsample_nearest_bwd(const void* dup, void* dx, int32_t B, int32_t H, int32_t W, int32_t Hup, int32_t Wup,
                              int32_t C, int32_t dtype, void* stream);
/* dy [B][Ho*Wo][C] -> z [B][H*W][C], z[2i][2j] = dy[i][j], 0 elsewhere */
int apad_zero_stuff2(const void* dy, void* z, int32_t B, int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t C,
                     int32_t dtype, void* stream);
/* x [M][C] -> xt [C][Mpad] zero padded (Mpad % 32 == 0) */
int apad_transpose_pad(const void* x, void* xt, int32_t M, int32_t C, int32_t Mpad, int32_t dtype, void* stream);
/* both operands of one weight-gradient GEMM in ONE launch: x0 [M][C0] -> xt0 [C0][Mpad], x1 [M][C1] -> xt1 [C1][Mpad] (columns >= M zero),
 * 16-bit inputs, outputs in the input type or widened to fp32 (out_f32 = 1: the exact-f32 MFMA path of the adapter gradients) (ABI 6) */
int apad_transpose_pad2(const void* x0, void* xt0, int32_t C0, const void* x1, void* xt1, int32_t C1, int32_t M, int32_t Mpad,
                        int32_t dtype, int32_t out_f32, void* stream);

/* fp32 workspace size of the three reductions below */
int64_t apad_reduce_workspace_bytes(void);
/* loss[0] = mean((pred - target)^2) in fp32 (train_apadapter_v2.py:954); dpred = grad_scale * 2 (pred - target) / n in dtype.
   grad_scale = the static loss scale of an f16 run (the reference's fp16 mode gets a GradScaler from accelerate), applied in
   fp32 before the rounding; 1 for bf16 */
int apad_mse_loss_grad(const void* pred, const float* target, void* dpred, float* loss, float* workspace, int64_t n,
                       float grad_scale, int32_t dtype, void* stream);
/* norm[0] = ||grad||_2 over the flat fp32 gradient buffer (clip_grad_norm_, :976) */
int apad_grad_norm(const float* grad, float* norm, float* workspace, int64_t n, void* stream);
/* step[0] += 1 unless grad_norm[0] is non-finite (NULL: always): device-side, so the step stays hipGraph-capturable */
int apad_step_advance_if_finite(int32_t* step, const float* grad_norm, void* stream);
/* (a non-finite grad_norm[0] -- f16 overflow under loss scaling -- makes the update a no-op)
   clip (coefficient min(1, max_norm / (norm + 1e-6)), read from device) + torch.optim.AdamW update (:763-769) of the
   flat fp32 master parameters; `work` (optional) receives the updated parameters in `dtype` for the forward kernels.
   step[0] = index of this step (>= 1), device int32. */
int apad_adamw_step(float* param, void* work, const float* grad, float* exp_avg, float* exp_avg_sq, const float* grad_norm,
                    const int32_t* step, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float max_grad_norm, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Audio front-end ("next" row f-2; audio_encoder/AudioMAE.py:356-394).  fp32.
 * ------------------------------------------------------------------------------------------------------------- */
/* torchaudio.functional.resample as a polyphase FIR: kernel [newf][2*width + orig] (sinc * hann^2, from the host),
   out[b*newf + p] = sum_j kernel[p][j] * xpad[b*orig + j], xpad = x zero-padded by `width` on the left */
int apad_resample_fir(const float* x, const float* kernel, float* out, int64_t n_in, int64_t n_out, int32_t orig,
                      int32_t newf, int32_t width, void* stream);
/* Kaldi fbank of x - dc at 16 kHz (25 ms hanning frames, 10 ms shift, snip_edges, per-frame DC removal, pre-emphasis,
   512-point power spectrum, mel banks [num_mel_bins][257], log) -> out [target_frames][num_mel_bins] =
   (logmel - norm_mean) / (2 norm_std), rows past the last frame = (0 - norm_mean) / (2 norm_std).
   window [400], twiddle [256][2] = (cos, -sin)(2 pi k / 512) */
int apad_kaldi_fbank(const float* x, int64_t n_samples, float dc, const float* window, const float* twiddle,
                     const float* mel, float* out, int32_t target_frames, int32_t num_mel_bins, float preemphasis,
                     float norm_mean, float norm_std, void* stream);

#ifdef __cplusplus
}
#endif
#endif
