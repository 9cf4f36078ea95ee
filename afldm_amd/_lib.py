"""ctypes binding of libafldm_hip.so (declarations mirror include/afldm_hip.h).

The library is the ONLY compute path of afldm_amd: if it cannot be loaded, importing this
module raises — there is no CPU fallback."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_size_t, c_void_p

import torch  # noqa: F401  (imported first so torch's libamdhip64.so.7 is the one HIP runtime in-process)

F32, BF16 = 0, 1
DTYPE_CODE = {torch.float32: F32, torch.bfloat16: BF16}

_HERE = os.path.dirname(os.path.abspath(__file__))
# AFLDM_LIB: alternative build of the same library (A/B timing of kernel changes on one GPU box)
LIB_PATH = os.environ.get("AFLDM_LIB") or os.path.join(_HERE, "lib", "libafldm_hip.so")


class ConvArgs(Structure):
    _fields_ = [
        ("x1", c_void_p), ("x2", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("temb", c_void_p),
        ("residual", c_void_p), ("y", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t),
        ("C1", c_int), ("C2", c_int), ("B", c_int), ("H", c_int), ("W", c_int), ("Cout", c_int),
        ("KS", c_int), ("temb_stride", c_int), ("res_ld", c_int), ("y_ld", c_int),
        ("out_mode", c_int), ("dtype", c_int), ("y2", c_void_p), ("split_n", c_int),
        ("stats_out", c_void_p), ("temb_mod", c_int), ("sync", c_void_p), ("sync_bytes", c_size_t),
        ("defer_reduce", c_int), ("w_batch_stride", ctypes.c_longlong),
        ("y_norm", c_void_p), ("norm_gamma", c_void_p), ("norm_beta", c_void_p), ("norm_groups", c_int), ("norm_eps", c_float),
        ("x_layout", c_int), ("y_layout", c_int),
    ]


class AfActArgs(Structure):
    """afldm_af_act_args (include/afldm_hip.h): the activation half of afldm_af_act_conv2d."""
    _fields_ = [
        ("x1", c_void_p), ("x2", c_void_p), ("C1", c_int), ("C2", c_int), ("stats1", c_void_p), ("S1", c_int),
        ("stats2", c_void_p), ("S2", c_int), ("gamma", c_void_p), ("beta", c_void_p), ("G", c_int), ("eps", c_float),
        ("U", c_void_p), ("D", c_void_p), ("packed", c_void_p),
    ]


class SepArgs(Structure):
    _fields_ = [
        ("x", c_void_p), ("y", c_void_p), ("M", c_void_p), ("M2", c_void_p), ("gn_table", c_void_p),
        ("outer_count", ctypes.c_longlong), ("inner_count", ctypes.c_longlong),
        ("in_outer_stride", ctypes.c_longlong), ("in_k_stride", ctypes.c_longlong),
        ("out_outer_stride", ctypes.c_longlong), ("out_k_stride", ctypes.c_longlong),
        ("K", c_int), ("R", c_int), ("R2", c_int), ("C", c_int), ("outer_per_sample", c_int),
        ("act", c_int), ("dtype", c_int), ("up_identity", c_int),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -m afldm_amd.build` (hipcc, gfx950). "
            "afldm_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    vp, ip, fp = c_void_p, c_int, c_float
    sigs = {
        "afldm_version": ([], c_int),
        "afldm_last_error": ([], c_char_p),
        "afldm_device_info": ([c_char_p, ip], c_int),
        "afldm_filter_matrix": ([ip, ip, ip, POINTER(c_float)], c_int),
        "afldm_nchw_to_nhwc": ([vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_nhwc_to_nchw": ([vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_pack_weight": ([vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_cast": ([vp, ip, vp, ip, c_size_t, vp], c_int),
        "afldm_timestep_embedding": ([vp, vp, ip, ip, ip, fp, ip, vp], c_int),
        "afldm_silu": ([vp, vp, c_size_t, ip, vp], c_int),
        "afldm_gn_stats_splits": ([ip], c_int),
        "afldm_gn_stats": ([vp, ip, vp, ip, ip, ip, vp], c_int),
        "afldm_gn_apply": ([vp, ip, vp, ip, vp, ip, vp, ip, vp, vp, vp, ip, ip, ip, fp, ip, ip, vp], c_int),
        "afldm_af_act": ([vp, ip, vp, ip, vp, ip, vp, ip, vp, vp, ip, fp, vp, vp, vp, vp, ip, ip, ip, vp], c_int),
        "afldm_af_act_c8": ([vp, ip, vp, ip, vp, ip, vp, ip, vp, vp, ip, fp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_af_act_const2": ([vp, ip, vp, ip, vp, ip, vp, ip, vp, vp, ip, fp, vp, vp, vp, ip, ip, vp], c_int),
        "afldm_conv2x2_const_norm_act_supported": ([ip, ip, ip, ip], c_int),
        "afldm_conv2x2_const_norm_act": ([vp, vp, vp, vp, ip, vp, vp, ip, fp, vp, vp, vp, ip, ip, ip, ip, vp], c_int),
        "afldm_af_act_slabs": ([vp, ip, vp, vp, ip, vp, vp, vp, vp, ip, fp, ip, vp, vp, vp, ip, ip, ip, ip, vp], c_int),
        "afldm_conv_out_fused": ([vp, vp, ip, vp, vp, ip, fp, vp, vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_af_pack_bytes": ([ip, ip], c_size_t),
        "afldm_af_pack": ([vp, vp, ip, ip, vp, vp], c_int),
        "afldm_af_up2": ([vp, vp, vp, vp, ip, ip, ip, ip, vp], c_int),
        "afldm_af_lpf_down2": ([vp, vp, vp, vp, vp, ip, ip, ip, ip, vp], c_int),
        "afldm_af_resample": ([vp, vp, vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_sep_pass": ([POINTER(SepArgs), vp], c_int),
        "afldm_af_resample_hw": ([vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_af_resample_plane": ([vp, vp, vp, vp, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_upfirdn2d": ([vp, vp, vp] + [ip] * 14 + [fp, ip, vp], c_int),
        "afldm_upfirdn2d_out_size": ([ip] * 6, c_int),
        "afldm_masked_metrics": ([vp, vp, vp, vp, ip, c_size_t, ip, vp], c_int),
        "afldm_gn_fold": ([vp, ip, vp, ip, ip, ip, vp], c_int),
        "afldm_gn_table": ([vp, ip, vp, vp, vp, ip, ip, ip, ip, fp, vp], c_int),
        "afldm_softmax_rows": ([vp, vp, ctypes.c_longlong, ip, fp, ip, vp], c_int),
        "afldm_conv2d": ([POINTER(ConvArgs), vp], c_int),
        "afldm_conv2d_workspace": ([POINTER(ConvArgs)], c_size_t),
        "afldm_conv2d_stats_splits": ([POINTER(ConvArgs)], c_int),
        "afldm_conv2d_tune": ([ip, ip], c_int),
        "afldm_conv2d_fused_splitk": ([ip], c_int),
        "afldm_conv2d_variant": ([POINTER(ConvArgs)], c_int),
        "afldm_conv2d_c8_ok": ([POINTER(ConvArgs)], c_int),
        "afldm_conv2d_norm_ok": ([POINTER(ConvArgs)], c_int),
        "afldm_attention": ([vp, ip, vp, ip, vp, vp, ip, ip, ip, ip, ip, ip, ip, fp, ip, vp], c_int),
        "afldm_attn_block_fused_supported": ([ip, ip, ip, ip], c_int),
        "afldm_attn_block_fused_trace": ([vp], c_int),
        "afldm_af_act_trace": ([vp], c_int),
        "afldm_attn_block_fused": ([vp, vp, ip, vp, vp, ip, fp, vp, vp, vp, ip, ip, ip, ip, fp, ip, vp], c_int),
        "afldm_attn_block_fused_out_supported": ([ip, ip, ip, ip, ip], c_int),
        "afldm_attn_block_fused_out": ([vp, vp, ip, vp, vp, ip, fp, vp, vp, vp, vp, vp, vp, vp, vp, c_longlong, ip, ip, ip, ip, fp, ip,
                                        vp], c_int),
        "afldm_ddim_step": ([vp, vp, vp, vp, vp, ip, ip, ip, ip, ip, ip, vp], c_int),
        "afldm_ddim_step_flat": ([vp, vp, vp, fp, fp, fp, fp, c_size_t, vp], c_int),
        "afldm_select_timestep": ([vp, vp, vp, ip, vp], c_int),
        "afldm_select_step_row": ([vp, vp, vp, ip, vp, vp, c_size_t, vp], c_int),
        "afldm_probe_mfma": ([vp, ip, ip, vp], c_int),
        "afldm_probe_mfma_random": ([vp, ip, ip, vp], c_int),
        "afldm_probe_copy": ([vp, vp, c_size_t, vp], c_int),
        "afldm_probe_chase": ([vp, vp, ip, vp], c_int),
        "afldm_probe_empty": ([ip, vp], c_int),
    }
    for name, (argtypes, restype) in sigs.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = restype
    return lib, sorted(sigs)


lib, EXPORTS = _load()


class AfldmError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib.afldm_last_error()
        raise AfldmError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def filter_matrix(kind: int, N: int, up: int = 2) -> torch.Tensor:
    """Host fp32 matrix: kind 0 -> U [up*N, N]; kind 1 -> D [N/2, N]; kind 2 -> L [N, N]."""
    rows, cols = {0: (up * N, N), 1: (N // 2, N), 2: (N, N)}[kind]
    buf = (c_float * (rows * cols))()
    check(lib.afldm_filter_matrix(kind, N, up, buf), "afldm_filter_matrix")
    return torch.frombuffer(buf, dtype=torch.float32).clone().reshape(rows, cols)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
