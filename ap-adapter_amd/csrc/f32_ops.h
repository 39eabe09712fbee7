// Internal entry points of the fp32 precision mode (f32_ops.hip).  The C-ABI functions dispatch here when the
// descriptor's dtype is APAD_F32; arguments are validated inside.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/apadapter_hip.h"

int apad_f32_gemm(const apad_gemm_desc* d, hipStream_t s);
int apad_f32_attention(const apad_attn_desc* d, hipStream_t s);
int apad_f32_layernorm(const void* x, const void* gamma, const void* beta, void* out, int64_t M, int32_t C, int64_t ldx,
                       int64_t ldo, float eps, hipStream_t s);
int apad_f32_groupnorm(const void* x, const void* gamma, const void* beta, void* out, int32_t B, int32_t HW, int32_t C,
                       int32_t G, float eps, int32_t silu, hipStream_t s);
int apad_f32_audiomae_pool(const void* rep, void* out, int32_t B, int32_t tp, int32_t fp, hipStream_t s);
int apad_f32_attention_bwd(const apad_attn_bwd_desc* d, hipStream_t s);
