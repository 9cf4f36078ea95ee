"""ctypes binding of libafldm_exp.so - the EXPERIMENTAL entry points of include/afldm_hip_experimental.h.

Not part of the product: merged activation / convolution launches, the cooperative 2x2-level trunk and the fused small-plane
attention were built, validated and measured slower (profiles/r05/); nothing on the default path imports this module's
library - it is loaded on first use by the opt-in switches (AFLDM_ACTCONV_N, AFLDM_TRUNK, AFLDM_ATTN_SMALL_T), the A/B tools
and the tests that keep the experiments correct."""
import ctypes
import os
from ctypes import POINTER, c_int, c_float, c_size_t, c_void_p

from ._lib import AfActArgs, ConvArgs, LIB_PATH

EXP_PATH = os.environ.get("AFLDM_EXP_LIB") or os.path.join(os.path.dirname(LIB_PATH), "libafldm_exp.so")
_state = {}


def available():
    return os.path.exists(EXP_PATH)


def lib():
    """The loaded experimental library (raises ImportError when it was not built)."""
    if "lib" not in _state:
        if not available():
            raise ImportError(f"{EXP_PATH} is missing: run `python -m afldm_amd.build` (the experiments are optional; the product "
                              "library does not need them)")
        h = ctypes.CDLL(EXP_PATH, mode=ctypes.RTLD_GLOBAL)
        vp, ip, fp = c_void_p, c_int, c_float
        sigs = {
            "afldm_af_act_conv2d_merged": ([POINTER(AfActArgs), POINTER(ConvArgs)], c_int),
            "afldm_af_act_conv2d": ([POINTER(AfActArgs), POINTER(ConvArgs), vp], c_int),
            "afldm_af_act_conv2d_trace": ([vp], c_int),
            "afldm_trunk_phase_bytes": ([], c_int),
            "afldm_trunk_trace": ([vp], c_int),
            "afldm_trunk_run": ([vp, ip, vp, vp, vp, ip, vp, vp, vp, c_size_t, vp], c_int),
            "afldm_act_conv_act_merged": ([POINTER(AfActArgs), POINTER(ConvArgs), POINTER(AfActArgs)], c_int),
            "afldm_act_conv_act": ([POINTER(AfActArgs), POINTER(ConvArgs), POINTER(AfActArgs), vp, vp], c_int),
            "afldm_af_act_conv2d_mode": ([ip], c_int),
            "afldm_attn_small_fused_supported": ([ip, ip, ip, ip], c_int),
            "afldm_attn_small_fused": ([vp, vp, vp, vp, ip, ip, ip, ip, fp, ip, vp], c_int),
        }
        for name, (argtypes, restype) in sigs.items():
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _state["lib"], _state["exports"] = h, sorted(sigs)
    return _state["lib"]


def exports():
    lib()
    return _state["exports"]
