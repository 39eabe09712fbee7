// Host-side plumbing of the C ABI: thread-local error message, launch check, descriptor self-checks.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include "common.h"

static thread_local char g_err[512] = "";

void apad_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int apad_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        apad_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

extern "C" const char* apad_last_error(void) { return g_err; }
int apad_ensure_dyn_lds(const void* kern, int bytes, unsigned* devmask) {
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 31) dev = 0;
    std::lock_guard<std::mutex> g(mu);
    if (!((*devmask >> dev) & 1u)) {
        const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) {
            apad_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) failed on device %d: %s", bytes, dev, hipGetErrorString(e));
            return -1;
        }
        *devmask |= 1u << dev;
    }
    return 0;
}

extern "C" int apad_abi_version(void) { return APAD_ABI_VERSION; }
extern "C" int apad_sizeof_gemm_desc(void) { return (int)sizeof(apad_gemm_desc); }
extern "C" int apad_sizeof_attn_desc(void) { return (int)sizeof(apad_attn_desc); }

#define PUT(x)                          \
    do {                                \
        if (n < cap) out[n] = (double)(x); \
        ++n;                            \
    } while (0)
#define PUTP(x) PUT((uintptr_t)(x))

extern "C" int apad_echo_gemm_desc(const apad_gemm_desc* d, double* out, int cap) {
    int n = 0;
    PUTP(d->a); PUTP(d->w); PUTP(d->out); PUTP(d->bias); PUTP(d->residual); PUTP(d->rowgroup_bias); PUTP(d->step_ptr);
    PUT(d->M); PUT(d->N); PUT(d->K); PUT(d->lda); PUT(d->ldw); PUT(d->ldo); PUT(d->ldr); PUT(d->ld_rg);
    PUT(d->rows_per_group);
    PUT(d->a_mode); PUT(d->epilogue); PUT(d->out_mode); PUT(d->dtype);
    PUT(d->Hin); PUT(d->Win); PUT(d->Cin); PUT(d->Hout); PUT(d->Wout); PUT(d->stride); PUT(d->Hup); PUT(d->Wup);
    PUT(d->src_batch_mod); PUT(d->residual_row_mod); PUT(d->heads); PUT(d->head_dim); PUT(d->L); PUT(d->Lpad);
    PUTP(d->out2); PUTP(d->out3);
    PUT(d->taps); PUT(d->dilation); PUT(d->pad); PUT(d->transposed); PUT(d->a_pre_act); PUT(d->a_pre_slope);
    PUT(d->conv_asym_pad); PUT(d->reserved_conv);
    PUTP(d->rowstat_out); PUTP(d->rowstat_in); PUTP(d->ln_colsum); PUTP(d->ln_bias); PUT(d->rowstat_in_tiles); PUT(d->ln_eps);
    PUTP(d->a2); PUT(d->lda2); PUT(d->k_split); PUT(d->a_row_mod); PUT(d->a2_row_mod); PUT(d->reserved_a2);
    PUTP(d->out4); PUTP(d->w_halo); PUTP(d->workspace); PUT(d->workspace_bytes);
    return n;
}

extern "C" int apad_echo_attn_desc(const apad_attn_desc* d, double* out, int cap) {
    int n = 0;
    PUTP(d->q); PUTP(d->k); PUTP(d->vt); PUTP(d->k2); PUTP(d->vt2); PUTP(d->out); PUTP(d->key_bias); PUTP(d->lse);
    PUT(d->q_stride_b); PUT(d->q_stride_n); PUT(d->k_stride_b); PUT(d->k_stride_l); PUT(d->vt_stride_b);
    PUT(d->k2_stride_b); PUT(d->k2_stride_l); PUT(d->vt2_stride_b); PUT(d->o_stride_b); PUT(d->o_stride_n);
    PUT(d->B); PUT(d->N); PUT(d->H); PUT(d->D); PUT(d->L); PUT(d->Lpad); PUT(d->L2); PUT(d->Lpad2);
    PUT(d->kv_batch_div); PUT(d->kv2_batch_div); PUT(d->dtype); PUT(d->softmax_scale); PUT(d->scale2); PUT(d->q_prescaled);
    return n;
}

extern "C" int apad_sizeof_rp_desc(void) { return (int)sizeof(apad_rp_desc); }
extern "C" int apad_echo_rp_desc(const apad_rp_desc* d, double* out, int cap) {
    int n = 0;
    PUTP(d->x); PUTP(d->w); PUTP(d->ln_gamma); PUTP(d->ln_beta); PUTP(d->residual);
    PUT(d->M); PUT(d->lda); PUT(d->ldw); PUT(d->ldr);
    PUT(d->K); PUT(d->epilogue); PUT(d->dtype); PUT(d->n_segments); PUT(d->ln_eps);
    PUT(d->heads); PUT(d->head_dim); PUT(d->L); PUT(d->Lpad);
    for (int i = 0; i < 3; ++i) {
        PUTP(d->seg[i].out); PUTP(d->seg[i].bias); PUT(d->seg[i].ldo); PUT(d->seg[i].n_cols); PUT(d->seg[i].mode);
    }
    return n;
}

extern "C" int apad_sizeof_mlp_desc(void) { return (int)sizeof(apad_mlp_desc); }
extern "C" int apad_echo_mlp_desc(const apad_mlp_desc* d, double* out, int cap) {
    int n = 0;
    PUTP(d->x); PUTP(d->ln_gamma); PUTP(d->ln_beta); PUTP(d->w1); PUTP(d->b1); PUTP(d->w2); PUTP(d->b2); PUTP(d->out);
    PUT(d->M); PUT(d->C); PUT(d->dtype); PUT(d->ln_eps); PUT(d->reserved);
    return n;
}

extern "C" int apad_sizeof_attn_bwd_desc(void) { return (int)sizeof(apad_attn_bwd_desc); }
extern "C" int apad_echo_attn_bwd_desc(const apad_attn_bwd_desc* d, double* out, int cap) {
    int n = 0;
    PUTP(d->q); PUTP(d->k); PUTP(d->v); PUTP(d->qt); PUTP(d->kt); PUTP(d->out); PUTP(d->dout); PUTP(d->doutt); PUTP(d->lse);
    PUTP(d->key_bias); PUTP(d->delta); PUTP(d->dq); PUTP(d->dk); PUTP(d->dv);
    PUT(d->B); PUT(d->N); PUT(d->H); PUT(d->D); PUT(d->L); PUT(d->Npad); PUT(d->Lpad); PUT(d->dtype);
    PUT(d->softmax_scale); PUT(d->dout_scale); PUT(d->accumulate_dq); PUT(d->ld_grad);
    return n;
}

extern "C" int apad_sizeof_xattn_desc(void) { return (int)sizeof(apad_xattn_desc); }
extern "C" int apad_sizeof_xrows_desc(void) { return (int)sizeof(apad_xrows_desc); }
extern "C" int apad_echo_xattn_desc(const apad_xattn_desc* d, double* out, int cap) {
    int n = 0;
    PUTP(d->x); PUTP(d->ln_gamma); PUTP(d->ln_beta); PUTP(d->wq_packed); PUTP(d->wo_packed); PUTP(d->bo); PUTP(d->kv1_packed);
    PUTP(d->key_bias); PUTP(d->kv2_packed); PUTP(d->out); PUTP(d->q_fold);
    PUT(d->B); PUT(d->N); PUT(d->C); PUT(d->heads); PUT(d->L1); PUT(d->L2);
    PUT(d->dtype); PUT(d->reserved); PUT(d->ln_eps); PUT(d->softmax_scale); PUT(d->scale2); PUT(d->reserved_f);
    return n;
}
