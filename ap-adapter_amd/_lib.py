"""Loader for libapadapter_hip.so (the C ABI declared in include/apadapter_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
torch is imported first so that the library binds to the HIP runtime PyTorch already loaded (same SONAME),
which is what makes torch streams / graph capture valid for our launches.
"""
import ctypes as C
import os
import subprocess
import threading

import torch  # noqa: F401  (must precede CDLL: shares libamdhip64 with PyTorch-ROCm)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APAD_LIB_PATH") or os.path.join(_HERE, "libapadapter_hip.so")  # override: kernel A/B experiments
CSRC = os.path.join(_HERE, "csrc")

BF16, F16, F32 = 0, 1, 2
A_PLAIN, A_CONV3X3, A_PATCH16, A_CONV1D = 0, 1, 2, 4
EPI_NONE, EPI_SILU, EPI_GELU, EPI_GEGLU, EPI_TANH, EPI_RELU, EPI_GELU_TANH, EPI_GEGLU_TANH = 0, 1, 2, 3, 4, 5, 6, 7
OUT_ROWMAJOR, OUT_VT, OUT_QKV = 0, 1, 2

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("a", "w", "out", "bias", "residual", "rowgroup_bias", "step_ptr")] + \
               [(n, _i64) for n in ("M", "N", "K", "lda", "ldw", "ldo", "ldr", "ld_rg", "rows_per_group")] + \
               [(n, _i32) for n in ("a_mode", "epilogue", "out_mode", "dtype", "Hin", "Win", "Cin", "Hout", "Wout",
                                    "stride", "Hup", "Wup", "src_batch_mod", "residual_row_mod", "heads", "head_dim", "L", "Lpad")] + \
               [("out2", _vp), ("out3", _vp)] + \
               [(n, _i32) for n in ("taps", "dilation", "pad", "transposed", "a_pre_act")] + [("a_pre_slope", _f32)] + \
               [("conv_asym_pad", _i32), ("reserved_conv", _i32)] + \
               [(n, _vp) for n in ("rowstat_out", "rowstat_in", "ln_colsum", "ln_bias")] + [("rowstat_in_tiles", _i32), ("ln_eps", _f32)] + \
               [("a2", _vp), ("lda2", _i64), ("k_split", _i32), ("a_row_mod", _i32), ("a2_row_mod", _i32), ("reserved_a2", _i32), ("out4", _vp),
                ("w_halo", _vp), ("workspace", _vp), ("workspace_bytes", _i64)]


class AttnDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("q", "k", "vt", "k2", "vt2", "out", "key_bias", "lse")] + \
               [(n, _i64) for n in ("q_stride_b", "q_stride_n", "k_stride_b", "k_stride_l", "vt_stride_b",
                                    "k2_stride_b", "k2_stride_l", "vt2_stride_b", "o_stride_b", "o_stride_n")] + \
               [(n, _i32) for n in ("B", "N", "H", "D", "L", "Lpad", "L2", "Lpad2", "kv_batch_div", "kv2_batch_div",
                                    "dtype")] + \
               [("softmax_scale", _f32), ("scale2", _f32), ("q_prescaled", _i32)]


class RpSegment(C.Structure):
    _fields_ = [("out", _vp), ("bias", _vp), ("ldo", _i64), ("n_cols", _i32), ("mode", _i32)]


class RpDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("x", "w", "ln_gamma", "ln_beta", "residual")] + \
               [(n, _i64) for n in ("M", "lda", "ldw", "ldr")] + \
               [(n, _i32) for n in ("K", "epilogue", "dtype", "n_segments")] + [("ln_eps", _f32)] + \
               [(n, _i32) for n in ("heads", "head_dim", "L", "Lpad")] + [("seg", RpSegment * 3)]


class MlpDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("x", "ln_gamma", "ln_beta", "w1", "b1", "w2", "b2", "out")] + \
               [("M", _i64), ("C", _i32), ("dtype", _i32), ("ln_eps", _f32), ("reserved", _i32)]


class XattnDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("x", "ln_gamma", "ln_beta", "wq_packed", "wo_packed", "bo", "kv1_packed", "key_bias", "kv2_packed", "out", "q_fold")] + \
               [(n, _i32) for n in ("B", "N", "C", "heads", "L1", "L2", "dtype", "reserved")] + \
               [(n, _f32) for n in ("ln_eps", "softmax_scale", "scale2", "reserved_f")]


class XrowsDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("x", "ln_gamma", "ln_beta", "wq_packed", "wo_packed", "bo", "k1", "vt1", "key_bias", "k2", "vt2", "out")] + \
               [(n, _i32) for n in ("B", "N", "C", "heads", "L1", "Lpad1", "L2", "Lpad2", "dtype", "reserved")] + \
               [(n, _f32) for n in ("ln_eps", "softmax_scale", "scale2", "reserved_f")]


class HsAttnDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("x", "w_packed", "w_bias", "k1", "vt1", "key_bias", "k2", "vt2", "out")] + \
               [(n, _i32) for n in ("B", "N", "C", "heads", "L1", "Lpad1", "L2", "Lpad2", "self_attention", "q_prescaled", "dtype", "normalize")] + \
               [(n, _f32) for n in ("ln_eps", "softmax_scale", "scale2", "reserved_f")]


class HsOutDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("o", "w_packed", "bias", "residual", "out", "rowstat_out")] + [(n, _i32) for n in ("B", "N", "C", "dtype")]


class AttnBwdDesc(C.Structure):
    _fields_ = [(n, _vp) for n in ("q", "k", "v", "qt", "kt", "out", "dout", "doutt", "lse", "key_bias", "delta", "dq", "dk", "dv")] + \
               [(n, _i32) for n in ("B", "N", "H", "D", "L", "Npad", "Lpad", "dtype")] + \
               [("softmax_scale", _f32), ("dout_scale", _f32), ("accumulate_dq", _i32), ("ld_grad", _i32)]


# name -> (restype, argtypes); every symbol include/apadapter_hip.h declares
SYMBOLS = {
    "apad_last_error": (C.c_char_p, []),
    "apad_abi_version": (C.c_int, []),
    "apad_set_gemm_ring": (C.c_int, [_i32]),
    "apad_sizeof_gemm_desc": (C.c_int, []),
    "apad_sizeof_attn_desc": (C.c_int, []),
    "apad_echo_gemm_desc": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(C.c_double), C.c_int]),
    "apad_echo_attn_desc": (C.c_int, [C.POINTER(AttnDesc), C.POINTER(C.c_double), C.c_int]),
    "apad_sizeof_rp_desc": (C.c_int, []),
    "apad_echo_rp_desc": (C.c_int, [C.POINTER(RpDesc), C.POINTER(C.c_double), C.c_int]),
    "apad_rowpanel_gemm": (C.c_int, [C.POINTER(RpDesc), _vp]),
    "apad_sizeof_mlp_desc": (C.c_int, []),
    "apad_echo_mlp_desc": (C.c_int, [C.POINTER(MlpDesc), C.POINTER(C.c_double), C.c_int]),
    "apad_geglu_mlp": (C.c_int, [C.POINTER(MlpDesc), _vp]),
    "apad_mlp_packed_bytes": (_i64, [_i32]),
    "apad_mlp_packed_bias_floats": (_i64, [_i32]),
    "apad_mlp_pack": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "apad_geglu_mlp_packed": (C.c_int, [C.POINTER(MlpDesc), _vp, _vp, _vp]),
    "apad_geglu_packed_bytes": (_i64, [_i32]),
    "apad_geglu_packed_bias_floats": (_i64, [_i32]),
    "apad_geglu_pack": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "apad_layernorm_geglu_packed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "apad_sizeof_xattn_desc": (C.c_int, []),
    "apad_echo_xattn_desc": (C.c_int, [C.POINTER(XattnDesc), C.POINTER(C.c_double), C.c_int]),
    "apad_fused_cross_attention": (C.c_int, [C.POINTER(XattnDesc), _vp]),
    "apad_sizeof_xrows_desc": (C.c_int, []),
    "apad_cross_attention_rows": (C.c_int, [C.POINTER(XrowsDesc), _vp]),
    "apad_sizeof_hs_attn_desc": (C.c_int, []),
    "apad_hs_attention": (C.c_int, [C.POINTER(HsAttnDesc), _vp]),
    "apad_self_attention_fused": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "apad_hs_geglu": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp]),
    "apad_sizeof_hs_out_desc": (C.c_int, []),
    "apad_hs_out": (C.c_int, [C.POINTER(HsOutDesc), _vp]),
    "apad_hs_ff2": (C.c_int, [C.POINTER(HsOutDesc), _vp]),
    "apad_xattn_pack_weight": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "apad_xattn_packed_kv_bytes": (_i64, [_i32, _i32]),
    "apad_xattn_pack_kv": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i64, _i64, _i64, _i32, _vp]),
    "apad_rows_packed_kv_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "apad_rows_pack_kv": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "apad_gemm": (C.c_int, [C.POINTER(GemmDesc), _vp]),
    "apad_conv_halo_pack": (C.c_int, [_vp, _vp, _i64, _i64, _i32, _vp]),
    "apad_hconv_launch_count": (_i64, []),
    "apad_probe_mfma": (C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_double), _vp]),
    "apad_conv_halo_packed_bytes": (_i64, [_i64, _i64]),
    "apad_conv_halo_workspace_bytes": (_i64, [_i64, _i64, _i64, _i32]),
    "apad_attention": (C.c_int, [C.POINTER(AttnDesc), _vp]),
    "apad_layernorm": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i64, _f32, _i32, _vp]),
    "apad_groupnorm_workspace_bytes": (_i64, [_i32, _i32, _i32]),
    "apad_groupnorm": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "apad_groupnorm2": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "apad_audiomae_pool": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "apad_timestep_embedding": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp]),
    "apad_cfg_ddim_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _i64, _i32, _vp]),
    "apad_step_advance": (C.c_int, [_vp, _vp]),
    "apad_mix3": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp]),
    "apad_softmax_rows": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _i64, _f32, _i32, _vp]),
    "apad_rmsnorm": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _f32, _i32, _i32, _vp]),
    "apad_gather_rows": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "apad_gaussian_sample": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    # training step (a-11)
    "apad_sizeof_attn_bwd_desc": (C.c_int, []),
    "apad_echo_attn_bwd_desc": (C.c_int, [C.POINTER(AttnBwdDesc), C.POINTER(C.c_double), C.c_int]),
    "apad_attention_bwd": (C.c_int, [C.POINTER(AttnBwdDesc), _vp]),
    "apad_head_transpose": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "apad_head_transpose3": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "apad_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "apad_layernorm_bwd_add": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _i32, _vp]),
    "apad_groupnorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp]),
    "apad_geglu": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "apad_geglu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "apad_upsample_nearest_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "apad_zero_stuff2": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "apad_transpose_pad": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "apad_transpose_pad2": (C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "apad_reduce_workspace_bytes": (_i64, []),
    "apad_mse_loss_grad": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _i32, _vp]),
    "apad_step_advance_if_finite": (C.c_int, [_vp, _vp, _vp]),
    "apad_grad_norm": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    # audio front-end (f-2)
    "apad_resample_fir": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "apad_kaldi_fbank": (C.c_int, [_vp, _i64, _f32, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _f32, _f32, _vp]),
    "apad_adamw_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _i32, _vp]),
}

_lock = threading.Lock()
_lib = None


def build(verbose=False):
    """Compile every HIP source for gfx950 into ap-adapter_amd/libapadapter_hip.so (hipcc cross-compiles; no GPU
    needed)."""
    r = subprocess.run(["make", "-C", CSRC, "-j4"], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:])
        print(r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("building libapadapter_hip.so failed:\n" + r.stderr[-2000:])
    return LIB_PATH


def lib():
    """The loaded library.  Raises RuntimeError when it is absent -- there is no CPU or PyTorch fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: the HIP extension is the product path and has no fallback; "
                    "run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C ap-adapter_amd/csrc`).")
            h = C.CDLL(LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(h, name)  # AttributeError if the ABI lost a symbol
                fn.restype = res
                fn.argtypes = args
            if h.apad_abi_version() != 10:
                raise RuntimeError("libapadapter_hip.so ABI version mismatch")
            if h.apad_sizeof_gemm_desc() != C.sizeof(GemmDesc) or h.apad_sizeof_attn_desc() != C.sizeof(AttnDesc) \
                    or h.apad_sizeof_rp_desc() != C.sizeof(RpDesc) \
                    or h.apad_sizeof_mlp_desc() != C.sizeof(MlpDesc) \
                    or h.apad_sizeof_attn_bwd_desc() != C.sizeof(AttnBwdDesc) \
                    or h.apad_sizeof_xattn_desc() != C.sizeof(XattnDesc) or h.apad_sizeof_xrows_desc() != C.sizeof(XrowsDesc) \
                    or h.apad_sizeof_hs_attn_desc() != C.sizeof(HsAttnDesc) or h.apad_sizeof_hs_out_desc() != C.sizeof(HsOutDesc):
                raise RuntimeError("descriptor layout mismatch between include/apadapter_hip.h and _lib.py")
            _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (rc={rc}): {lib().apad_last_error().decode()}")
