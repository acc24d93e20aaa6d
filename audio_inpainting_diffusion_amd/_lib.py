"""ctypes binding of libaid_hip.so (the C ABI in include/aid_kernels.h).

There is deliberately NO fallback: if the shared library is missing or a launcher returns an error code,
``AidError`` is raised.  Tensors are passed as raw device pointers + element strides; the stream is torch's
current HIP stream, so kernels order naturally with torch allocations/copies and can be graph-captured.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (AID_LIB_PATH selects another build of the library for kernel A/B experiments -- honoured only together with AID_EXPERIMENT=1, and announced on
#  stderr, so that a stray variable can never silently change what a run measures)
LIB_PATH = os.path.join(_HERE, "libaid_hip.so")
if os.environ.get("AID_EXPERIMENT") == "1" and os.environ.get("AID_LIB_PATH"):
    LIB_PATH = os.environ["AID_LIB_PATH"]
    import sys as _sys
    print(f"[aid] AID_EXPERIMENT=1: loading {LIB_PATH} instead of the in-tree libaid_hip.so", file=_sys.stderr)
AID_CQT_MAX_OCT = 12
AID_STATS_SPLIT = 32
AID_CONV2D_SPLIT_FLAG_BYTES = 4096     # include/aid_kernels.h: leading flag region of the split-K scratch (checked against the header in tests/test_host_logic.py)
AID_ATTN_MAX_T = 128          # aid_time_attention: one workgroup holds a whole [T, T] score tile


class AidError(RuntimeError):
    pass


class View(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sB", C.c_int64), ("sC", C.c_int64), ("sF", C.c_int64)]


class GroupStatsParams(C.Structure):
    _fields_ = [("x", View), ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int), ("groups", C.c_int),
                ("gamma", C.c_void_p), ("mod", C.c_void_p), ("mod_ld", C.c_int64), ("eps", C.c_float),
                ("scale", C.c_void_p), ("stats", C.c_void_p), ("ws", C.c_void_p), ("ws_n", C.c_int)]


class Conv2dParams(C.Structure):
    _fields_ = [("x", View), ("y", View), ("res", View), ("aux", View),
                ("wp", C.c_void_p),
                ("in_scale", C.c_void_p), ("in_scale_ld", C.c_int64),
                ("out_scale", C.c_void_p), ("out_scale_ld", C.c_int64),
                ("aux_scale", C.c_void_p), ("aux_scale_ld", C.c_int64),
                ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("F", C.c_int), ("T", C.c_int),
                ("Cin_pad", C.c_int), ("Cout_pad", C.c_int),
                ("KH", C.c_int), ("KW", C.c_int), ("dilF", C.c_int),
                ("act", C.c_int), ("epi", C.c_int),
                ("alpha", C.c_float), ("res_scale", C.c_float), ("wp_wino", C.c_void_p), ("wino_taps", C.c_int),
                ("x_wino", C.c_int), ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
                ("dot_ws", C.c_void_p), ("dot_n", C.c_int), ("stat_ws", C.c_void_p), ("stat_n", C.c_int),
                ("x2", View), ("Cin1", C.c_int),
                ("fin_mode", C.c_int), ("fin_count", C.c_void_p), ("fin_gamma", C.c_void_p), ("fin_mod", C.c_void_p), ("fin_mod_ld", C.c_int64),
                ("fin_eps", C.c_float), ("fin_scale", C.c_void_p), ("fin_stats", C.c_void_p)]


class Wino2dGemmParams(C.Structure):
    _fields_ = [("U", C.c_void_p), ("V", C.c_void_p), ("M", C.c_void_p), ("nxi", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("Cin_pad", C.c_int), ("Cout_pad", C.c_int), ("N", C.c_int64), ("variant", C.c_int)]


class ResampleParams(C.Structure):
    _fields_ = [("x", View), ("y", View), ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int),
                ("up", C.c_int), ("adjoint", C.c_int), ("accumulate", C.c_int)]


class AttentionParams(C.Structure):
    _fields_ = [("qk", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p), ("probs", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("F", C.c_int), ("T", C.c_int), ("scale", C.c_float), ("bias", C.c_void_p)]


class EmbedParams(C.Structure):
    _fields_ = [("sigma", C.c_void_p), ("rff_freq", C.c_void_p), ("w0", C.c_void_p), ("b0", C.c_void_p),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("emb", C.c_void_p),
                ("B", C.c_int), ("rff", C.c_int), ("h0", C.c_int), ("h1", C.c_int), ("E", C.c_int)]


class ModulationParams(C.Structure):
    _fields_ = [("emb", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("mod", C.c_void_p),
                ("B", C.c_int), ("E", C.c_int), ("N", C.c_int)]


class CqtTables(C.Structure):
    _fields_ = [("n_oct", C.c_int), ("bins", C.c_int), ("rc", C.c_void_p), ("Lg", C.c_void_p), ("goff", C.c_void_p),
                ("g", C.c_void_p), ("T_oct", C.c_void_p), ("twiddle", C.c_void_p), ("Tmax", C.c_int)]


class CqtParams(C.Structure):
    _fields_ = [("tab", CqtTables), ("T_host", C.c_int * AID_CQT_MAX_OCT), ("oct", View * AID_CQT_MAX_OCT),
                ("spec", C.c_void_p), ("band_ws", C.c_void_p), ("in_scale", C.c_void_p), ("B", C.c_int), ("Lh", C.c_int),
                ("unnormalized", C.c_int)]


class CqtGatherParams(C.Structure):
    _fields_ = [("band_ws", C.c_void_p), ("kfirst", C.c_void_p), ("kcount", C.c_void_p),
                ("rc", C.c_void_p), ("Lg", C.c_void_p), ("goff", C.c_void_p), ("woff", C.c_void_p), ("Tk", C.c_void_p),
                ("gdM", C.c_void_p), ("X", C.c_void_p), ("cskip", C.c_void_p), ("cout", C.c_void_p), ("hpf", C.c_void_p),
                ("Y", C.c_void_p), ("B", C.c_int), ("Lh", C.c_int), ("ws_per_b", C.c_int64), ("band_scale", C.c_void_p)]


class AxpbyParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("out", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p),
                ("B", C.c_int), ("L", C.c_int64), ("a_host", C.c_float), ("b_host", C.c_float)]

    def __init__(self, x, y, out, a, b, B, L, a_host=1.0, b_host=1.0):
        super().__init__(x, y, out, a, b, B, L, a_host, b_host)


class ScoreStepParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("xhat", C.c_void_p), ("yobs", C.c_void_p), ("smask", C.c_void_p),
                ("smask_sB", C.c_int64), ("x0", C.c_void_p), ("d0", C.c_void_p), ("t", C.c_void_p), ("h", C.c_void_p),
                ("xnext", C.c_void_p), ("dout", C.c_void_p), ("xh_out", C.c_void_p),
                ("B", C.c_int), ("L", C.c_int64), ("mode", C.c_int), ("t_host", C.c_float), ("h_host", C.c_float)]


class Add2Params(C.Structure):
    _fields_ = [("u", View), ("v", View), ("y", View), ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int),
                ("a", C.c_float), ("b", C.c_float)]


class ScaleActParams(C.Structure):
    _fields_ = [("x", View), ("y", View), ("scale", C.c_void_p), ("scale_ld", C.c_int64),
                ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int), ("act", C.c_int), ("wino", C.c_int), ("dilF", C.c_int), ("mul", C.c_float)]


class FftPassParams(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("twiddle", C.c_void_p), ("B", C.c_int), ("N", C.c_int), ("R", C.c_int),
                ("Ns", C.c_int), ("in_mode", C.c_int), ("out_mode", C.c_int), ("sign", C.c_float), ("out_scale", C.c_float)]


class GroupDotParams(C.Structure):
    _fields_ = [("u", View), ("v", View), ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int), ("groups", C.c_int),
                ("ws", C.c_void_p)]


class NormBwdParams(C.Structure):
    _fields_ = [("gd", View), ("x", View), ("gy", View), ("out", View),
                ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int), ("groups", C.c_int),
                ("stats", C.c_void_p), ("ws", C.c_void_p), ("eps", C.c_float), ("a", C.c_float), ("accumulate", C.c_int),
                ("ws_n", C.c_int), ("wout", View), ("wscale", C.c_void_p), ("wscale_ld", C.c_int64), ("wform", C.c_int), ("coef_ready", C.c_int), ("wdil", C.c_int)]


class AttentionBwdParams(C.Structure):
    _fields_ = [("qk", C.c_void_p), ("v", C.c_void_p), ("probs", C.c_void_p), ("gout", C.c_void_p),
                ("gqk", C.c_void_p), ("gv", C.c_void_p), ("B", C.c_int), ("H", C.c_int), ("F", C.c_int), ("T", C.c_int),
                ("scale", C.c_float), ("accumulate_gv", C.c_int), ("ws", C.c_void_p)]


class GuidanceSeedParams(C.Structure):
    _fields_ = [("xhat", C.c_void_p), ("y", C.c_void_p), ("mask", C.c_void_p), ("mask_sB", C.c_int64),
                ("g", C.c_void_p), ("norm", C.c_void_p), ("B", C.c_int), ("L", C.c_int64), ("norm_type", C.c_int), ("beta", C.c_float)]

    def __init__(self, xhat, y, mask, mask_sB, g, norm, B, L, norm_type=2, beta=1.0):
        super().__init__(xhat, y, mask, mask_sB, g, norm, B, L, norm_type, beta)


class GuidanceStepParams(C.Structure):
    _fields_ = [("xhat", C.c_void_p), ("g", C.c_void_p), ("out", C.c_void_p), ("step_out", C.c_void_p), ("s_out", C.c_void_p),
                ("B", C.c_int), ("L", C.c_int64), ("coef", C.c_float), ("inv_sqrt_len", C.c_float), ("eps", C.c_float)]


class SetRowsParams(C.Structure):
    _fields_ = [("out", C.c_void_p), ("ld", C.c_int64), ("B", C.c_int), ("n", C.c_int), ("v", C.c_float * 8)]


class RowNormParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("B", C.c_int), ("L", C.c_int64)]


class ResamplePolyParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("kernel", C.c_void_p), ("x_ld", C.c_int64), ("y_ld", C.c_int64),
                ("L", C.c_int64), ("Lout", C.c_int64), ("B", C.c_int), ("orig_freq", C.c_int), ("new_freq", C.c_int),
                ("width", C.c_int), ("K", C.c_int)]


class StftParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("frames", C.c_void_p), ("out", C.c_void_p),
                ("window", C.c_void_p), ("twiddle", C.c_void_p), ("inv_env", C.c_void_p),
                ("mask", C.c_void_p), ("mask_sB", C.c_int64), ("mask_ld", C.c_int64),
                ("add1", C.c_void_p), ("add2", C.c_void_p), ("c0", C.c_float),
                ("B", C.c_int), ("L", C.c_int64), ("Lp", C.c_int64),
                ("n_fft", C.c_int), ("hop", C.c_int), ("n_frames", C.c_int), ("adjoint", C.c_int)]


class WgradParams(C.Structure):
    _fields_ = [("gy", View), ("x", View), ("P", C.c_void_p),
                ("B", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("F", C.c_int), ("T", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
                ("dilF", C.c_int), ("S", C.c_int), ("alpha", C.c_float), ("wino", C.c_int)]


class WinoGyParams(C.Structure):
    _fields_ = [("gy", View), ("out", View), ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int)]


class PackConvWeightParams(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("wpT", C.c_void_p), ("wpw", C.c_void_p), ("wpwT", C.c_void_p),
                ("Cout", C.c_int), ("Cin", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
                ("Cin_pad", C.c_int), ("Cout_pad", C.c_int), ("Cin_padT", C.c_int), ("Cout_padT", C.c_int),
                ("wpw8", C.c_void_p), ("wpw8T", C.c_void_p), ("wpw2", C.c_void_p), ("wpw2T", C.c_void_p), ("wpw3", C.c_void_p), ("wpw3T", C.c_void_p)]


class WgradReduceParams(C.Structure):
    _fields_ = [("P", C.c_void_p), ("W", C.c_void_p), ("gate", C.c_void_p), ("gate_ld", C.c_int64),
                ("in_scale", C.c_void_p), ("in_scale_ld", C.c_int64), ("dW", C.c_void_p), ("dgate", C.c_void_p), ("dgate_ld", C.c_int64),
                ("B", C.c_int), ("S", C.c_int), ("Cout", C.c_int), ("Cin", C.c_int), ("K", C.c_int), ("accumulate", C.c_int),
                ("wino", C.c_int), ("Uw", C.c_void_p), ("Cin_pad", C.c_int), ("Cout_pad", C.c_int)]


class ChannelDotParams(C.Structure):
    _fields_ = [("u", View), ("v", View), ("out", C.c_void_p), ("out_ld", C.c_int64),
                ("B", C.c_int), ("C", C.c_int), ("F", C.c_int), ("T", C.c_int)]


class RelposBwdParams(C.Structure):
    _fields_ = [("dS", C.c_void_p), ("bucket", C.c_void_p), ("dW", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("T", C.c_int), ("num_buckets", C.c_int), ("accumulate", C.c_int)]


class ScaleBwdParams(C.Structure):
    _fields_ = [("S", C.c_void_p), ("S_ld", C.c_int64), ("scale", C.c_void_p), ("scale_ld", C.c_int64), ("gamma", C.c_void_p),
                ("mod", C.c_void_p), ("mod_ld", C.c_int64), ("stats", C.c_void_p), ("dgamma", C.c_void_p), ("dmod", C.c_void_p),
                ("dmod_ld", C.c_int64), ("B", C.c_int), ("C", C.c_int), ("groups", C.c_int), ("accumulate", C.c_int)]


class ModulationBwdParams(C.Structure):
    _fields_ = [("dmod", C.c_void_p), ("emb", C.c_void_p), ("W", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p),
                ("demb", C.c_void_p), ("B", C.c_int), ("E", C.c_int), ("N", C.c_int), ("accumulate", C.c_int),
                ("part", C.c_void_p), ("part_floats", C.c_int64)]


class EmbedBwdParams(C.Structure):
    _fields_ = [("fwd", EmbedParams), ("demb", C.c_void_p), ("dw0", C.c_void_p), ("db0", C.c_void_p), ("dw1", C.c_void_p),
                ("db1", C.c_void_p), ("dw2", C.c_void_p), ("db2", C.c_void_p), ("accumulate", C.c_int)]


class AdamParams(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("gscale", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bias1", C.c_float), ("bias2_sqrt", C.c_float)]


class EmaParams(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("n", C.c_int64), ("rate", C.c_float)]


class SumsqParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ws", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("max_norm", C.c_float)]


AID_SUMSQ_BLOCKS = 512

EXPORTS = ["aid_abi_version", "aid_last_error", "aid_last_kernel", "aid_group_stats", "aid_conv2d", "aid_conv2d_pack_dims", "aid_conv2d_stat_partials", "aid_pack_conv_weight", "aid_wino_gy", "aid_conv2d_wgrad_tiles", "aid_resample",
           "aid_time_attention", "aid_embed", "aid_modulation", "aid_cqt_analysis", "aid_cqt_synthesis",
           "aid_cqt_gather", "aid_axpby", "aid_score_step", "aid_add2", "aid_group_dot", "aid_norm_bwd",
           "aid_time_attention_bwd", "aid_guidance_seed", "aid_guidance_step", "aid_set_rows", "aid_row_norm", "aid_scale_act", "aid_fft_pass",
           "aid_stft_frames", "aid_stft_ola", "aid_resample_poly", "aid_conv2d_wino_input_supported", "aid_conv2d_dot_partials", "aid_conv2d_x2_supported", "aid_conv2d_dot_partials_1x1", "aid_conv2d_wino_input_ok", "aid_conv2d_wino_form", "aid_conv2d_wino8_supported", "aid_conv2d_wino_split_ws_bytes", "aid_conv2d_fin_supported",
           "aid_conv2d_wgrad", "aid_wgrad_reduce", "aid_channel_dot", "aid_relpos_bwd", "aid_scale_bwd", "aid_modulation_bwd", "aid_embed_bwd",
           "aid_adam", "aid_ema", "aid_sumsq", "aid_wino2d_gemm", "aid_wino2d_set_split", "aid_conv2d_wino2d_gemm", "aid_conv2d_wino2d_output", "aid_conv2d_wino2d_supported", "aid_conv2d_wino2d_positions", "aid_conv2d_wino2d_wanted", "aid_conv2d_wino2d_tform"]

_lib = None


def lib():
    """Load libaid_hip.so (built by ``__graft_entry__.build()`` / ``build.py``).  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AidError(f"{LIB_PATH} not found: build the HIP extension first (python __graft_entry__.py). "
                           "There is no CPU / eager fallback on the product path.")
        L = C.CDLL(LIB_PATH)
        L.aid_last_error.restype = C.c_char_p
        L.aid_abi_version.restype = C.c_int
        L.aid_last_kernel.restype = C.c_char_p
        for name in EXPORTS:
            if not hasattr(L, name):
                raise AidError(f"libaid_hip.so does not export {name}")
        L.aid_conv2d_pack_dims.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.aid_conv2d_pack_dims.restype = None
        L.aid_conv2d_wino_input_supported.argtypes = [C.c_int, C.c_int, C.c_int]
        L.aid_conv2d_wino_input_supported.restype = C.c_int
        L.aid_conv2d_dot_partials.argtypes = [C.c_int] * 7
        L.aid_conv2d_dot_partials.restype = C.c_int
        L.aid_conv2d_x2_supported.argtypes = [C.c_int] * 5
        L.aid_conv2d_x2_supported.restype = C.c_int
        L.aid_conv2d_dot_partials_1x1.argtypes = [C.c_int] * 5
        L.aid_conv2d_dot_partials_1x1.restype = C.c_int
        L.aid_conv2d_wino_input_ok.argtypes = [C.c_int] * 6
        L.aid_conv2d_wino_input_ok.restype = C.c_int
        L.aid_conv2d_wino8_supported.argtypes = [C.c_int] * 5
        L.aid_conv2d_wino8_supported.restype = C.c_int
        L.aid_conv2d_fin_supported.argtypes = [C.c_int] * 7
        L.aid_conv2d_fin_supported.restype = C.c_int
        L.aid_conv2d_wino_form.argtypes = [C.c_int] * 6
        L.aid_conv2d_wino_form.restype = C.c_int
        L.aid_conv2d_wino_split_ws_bytes.argtypes = [C.c_int] * 6
        L.aid_conv2d_wino_split_ws_bytes.restype = C.c_int64
        L.aid_conv2d_stat_partials.argtypes = [C.c_int] * 7
        L.aid_conv2d_stat_partials.restype = C.c_int
        L.aid_conv2d_wgrad_tiles.argtypes = [C.c_int] * 5
        L.aid_conv2d_wgrad_tiles.restype = C.c_int
        L.aid_conv2d_wino2d_supported.argtypes = [C.c_int] * 5
        L.aid_conv2d_wino2d_supported.restype = C.c_int
        L.aid_conv2d_wino2d_wanted.argtypes = [C.c_int] * 6
        L.aid_conv2d_wino2d_wanted.restype = C.c_int
        L.aid_conv2d_wino2d_tform.argtypes = [C.c_int] * 6
        L.aid_conv2d_wino2d_tform.restype = C.c_int
        L.aid_conv2d_wino2d_positions.argtypes = [C.c_int] * 4
        L.aid_conv2d_wino2d_positions.restype = C.c_int64
        L.aid_wino2d_set_split.argtypes = [C.c_int]
        L.aid_wino2d_set_split.restype = C.c_int
        for name in EXPORTS[3:]:
            if name not in ("aid_conv2d_pack_dims", "aid_conv2d_wino_input_supported", "aid_conv2d_dot_partials", "aid_conv2d_x2_supported", "aid_conv2d_dot_partials_1x1", "aid_conv2d_wino_input_ok", "aid_conv2d_wino_form", "aid_conv2d_wino8_supported", "aid_conv2d_wino_split_ws_bytes", "aid_conv2d_fin_supported",
                            "aid_conv2d_stat_partials", "aid_conv2d_wgrad_tiles", "aid_conv2d_wino2d_supported", "aid_conv2d_wino2d_positions", "aid_conv2d_wino2d_wanted", "aid_conv2d_wino2d_tform", "aid_wino2d_set_split"):
                getattr(L, name).argtypes = [C.c_void_p, C.c_void_p]
                getattr(L, name).restype = C.c_int
        if L.aid_abi_version() != 14:
            raise AidError("ABI version mismatch")
        _lib = L
    return _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, params) -> None:
    rc = getattr(lib(), name)(C.addressof(params), _stream())
    if rc != 0:
        raise AidError(f"{name} failed rc={rc}: {lib().aid_last_error().decode()}")


def call_on(fn, params_addr: int, stream: int, name: str = "") -> None:
    rc = fn(params_addr, stream)
    if rc != 0:
        raise AidError(f"{name} failed rc={rc}: {lib().aid_last_error().decode()}")


def _req(t: torch.Tensor):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise AidError("HIP kernels need float32 tensors on the GPU (no CPU fallback)")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def view4(t: Optional[torch.Tensor]) -> View:
    """[B,C,F,T] tensor (any strides with T contiguous) -> aid_view."""
    if t is None:
        return View(None, 0, 0, 0)
    _req(t)
    assert t.dim() == 4 and (t.stride(3) == 1 or t.shape[3] == 1), "T must be contiguous"
    return View(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def pack_dims(cin: int, cout: int):
    a, b = C.c_int(0), C.c_int(0)
    lib().aid_conv2d_pack_dims(cin, cout, C.byref(a), C.byref(b))
    return a.value, b.value


def pack_conv_weight(w: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """[Cout,Cin,KH,KW] (or [Cout,Cin,1] / [Cout,Cin]) -> packed [KH*KW, Cin_pad, Cout_pad] (cout contiguous).

    transpose=True packs the input-gradient operator instead: taps flipped, roles of Cin/Cout swapped."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    elif w.dim() == 3:
        w = w[:, :, None, :]  # Conv1d [O,I,k] -> [O,I,1,k]
    w = w.detach().float()
    if transpose:
        w = w.flip(2, 3).permute(1, 0, 2, 3)
    co, ci, kh, kw = w.shape
    cip, cop = pack_dims(ci, co)
    out = torch.zeros(kh * kw, cip, cop, device=w.device, dtype=torch.float32)
    out[:, :ci, :co] = w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co)
    return out


def pack_conv_weight_wino(w: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """[Cout,Cin,5,3] -> Winograd F(4,3) pack [6*5, Cin_pad, Cout_pad]: U = G w along kw, "tap" index xi*5+kh
    (computed in float64, stored fp32)."""
    w = w.detach().double()
    if transpose:
        w = w.flip(2, 3).permute(1, 0, 2, 3)
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (5, 3)
    w0, w1, w2 = w[..., 0], w[..., 1], w[..., 2]
    U = torch.stack((w0 / 4, -(w0 + w1 + w2) / 6, -(w0 - w1 + w2) / 6, (w0 + 2 * w1 + 4 * w2) / 24,
                     (w0 - 2 * w1 + 4 * w2) / 24, w2), dim=0)                              # [NXI, co, ci, kh]
    nxi = U.shape[0]
    cip, cop = pack_dims(ci, co)
    out = torch.zeros(nxi * kh, cip, cop, device=w.device, dtype=torch.float32)
    out[:, :ci, :co] = U.permute(0, 3, 2, 1).reshape(nxi * kh, ci, co).float()
    return out


W8_POINTS = (0.4, 0.8, 1.25, 2.5)


def wino8_matrices():
    """(A^T [8,10], G [10,3], B^T [10,10]) of F(8,3) as the kernels use them (float64; same construction as tools/gen_wino8.py, which
    writes csrc/aid_wino8.h): points {0, +-0.4, +-0.8, +-1.25, +-2.5, inf}, rows of B^T scaled to max |entry| = 1 and G by the inverse."""
    import numpy as np
    n = 10
    pts = np.array([0.0] + [sg * a for a in W8_POINTS for sg in (1, -1)], dtype=np.float64)

    def vander(cols):
        V = np.zeros((n, cols))
        for j in range(n - 1):
            V[j] = pts[j] ** np.arange(cols)
        V[n - 1, cols - 1] = 1.0
        return V
    AT, G, BT = vander(8).T, vander(3), np.linalg.inv(vander(n)).T
    sc = np.abs(BT).max(axis=1)
    BT = BT / sc[:, None]
    BT[np.abs(BT) < 1e-13] = 0.0
    return AT, G * sc[:, None], BT


def wino45_matrices():
    """(AF^T [4,8], GF [8,5], BF^T [8,8], AT^T [4,6], GT [6,3], BT^T [6,6]) of the 2-D form F(4,5) x F(4,3) as the kernels use them (float64; same
    construction as tools/gen_wino45.py, which writes csrc/aid_wino45.h): row-axis points {0, +-1, +-2, +-1/2, inf}, T-axis {0, +-1, +-2, inf}."""
    import numpy as np

    def toom(points, m, r):
        n = m + r - 1
        pts = np.array([0.0] + [sg * a for a in points for sg in (1, -1)], dtype=np.float64)

        def vander(cols):
            V = np.zeros((n, cols))
            for j in range(n - 1):
                V[j] = pts[j] ** np.arange(cols)
            V[n - 1, cols - 1] = 1.0
            return V
        AT, G, BT = vander(m).T, vander(r), np.linalg.inv(vander(n)).T
        sc = np.abs(BT).max(axis=1)
        BT = BT / sc[:, None]
        BT[np.abs(BT) < 1e-13] = 0.0
        return AT, G * sc[:, None], BT
    return toom((1.0, 2.0, 0.5), 4, 5) + toom((1.0, 2.0), 4, 3)


def pack_conv_weight_wino2d(w: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """[Cout,Cin,5,3] -> 2-D Winograd pack [48, Cin_pad, Cout_pad]: U[xf*6 + xt] = GF w GT^T (float64, stored fp32)."""
    w = w.detach().double()
    if transpose:
        w = w.flip(2, 3).permute(1, 0, 2, 3)
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (5, 3)
    m = wino45_matrices()
    GF, GT = torch.from_numpy(m[1]).to(w.device), torch.from_numpy(m[4]).to(w.device)
    U = torch.einsum("fh,tk,oihk->ftio", GF, GT, w).reshape(48, ci, co)
    cip, cop = pack_dims(ci, co)
    out = torch.zeros(48, cip, cop, device=w.device, dtype=torch.float32)
    out[:, :ci, :co] = U.float()
    return out


def pack_conv_weight_wino2d8(w: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """[Cout,Cin,5,3] -> pack of the 2-D form with F(8,3) along T [80, Cin_pad, Cout_pad]: U[xf*10 + xt] = GF w G8^T (float64, stored fp32)."""
    w = w.detach().double()
    if transpose:
        w = w.flip(2, 3).permute(1, 0, 2, 3)
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (5, 3)
    GF = torch.from_numpy(wino45_matrices()[1]).to(w.device)
    G8 = torch.from_numpy(wino8_matrices()[1]).to(w.device)
    U = torch.einsum("fh,tk,oihk->ftio", GF, G8, w).reshape(80, ci, co)
    cip, cop = pack_dims(ci, co)
    out = torch.zeros(80, cip, cop, device=w.device, dtype=torch.float32)
    out[:, :ci, :co] = U.float()
    return out


def pack_conv_weight_wino8(w: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """[Cout,Cin,5,3] -> Winograd F(8,3) pack [10*5, Cin_pad, Cout_pad]: U = G w along kw, "tap" index xi*5+kh (float64, stored fp32)."""
    w = w.detach().double()
    if transpose:
        w = w.flip(2, 3).permute(1, 0, 2, 3)
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (5, 3)
    G = torch.from_numpy(wino8_matrices()[1]).to(w.device)
    U = torch.einsum("xk,oihk->xoih", G, w)                                               # [NXI, co, ci, kh]
    cip, cop = pack_dims(ci, co)
    out = torch.zeros(10 * kh, cip, cop, device=w.device, dtype=torch.float32)
    out[:, :ci, :co] = U.permute(0, 3, 2, 1).reshape(10 * kh, ci, co).float()
    return out


# ---------------------------------------------------------------------------------------------------------
# CQT launch helpers (used by cqt.CQTransform)
# ---------------------------------------------------------------------------------------------------------
def _cqt_params(tr, tab, octs: Sequence[torch.Tensor], window: Optional[torch.Tensor] = None) -> CqtParams:
    P = tr.plan
    p = CqtParams()
    p.tab = CqtTables(P.numocts, P.binsoct, ptr(tab["rc"]), ptr(tab["Lg"]), ptr(tab["goff"]),
                      ptr(tab["g"] if window is None else window), ptr(tab["T_oct"]), ptr(tab["twiddle"]), P.Tmax)
    for o in range(P.numocts):
        p.T_host[o] = int(P.T_oct[o])
        t = octs[o]
        _req(t)
        assert t.shape[1] == 2 and t.shape[2] == P.binsoct and t.shape[3] == P.T_oct[o] and t.stride(3) == 1
        p.oct[o] = View(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))
    p.B = octs[0].shape[0]
    p.Lh = P.Lh
    p.unnormalized = 0
    return p


def cqt_analysis(tr, tab, spec_ri: torch.Tensor, octs, in_scale=None, window=None, unnormalized=False):
    """spec [B,Lh,2] -> planar octave views.  window=None: analysis windows g (forward transform);
    window=gdM + unnormalized: exact adjoint of synthesis+gather (input-VJP)."""
    _req(spec_ri)
    assert spec_ri.is_contiguous()
    p = _cqt_params(tr, tab, octs, window)
    p.spec = spec_ri.data_ptr()
    p.in_scale = ptr(in_scale)
    p.unnormalized = int(unnormalized)
    call("aid_cqt_analysis", p)


def cqt_synthesis(tr, tab, octs, ws: torch.Tensor):
    p = _cqt_params(tr, tab, octs)
    p.band_ws = ws.data_ptr()
    call("aid_cqt_synthesis", p)


def cqt_gather(tr, tab, ws, Y, X=None, cskip=None, cout=None, hpf=None, window=None, band_scale=None):
    """Y = hpf * (cskip*X + cout * band_scale * sum_k ws_k * window_k).  window=None: dual windows gdM (forward
    synthesis); window = g/T_k: adjoint of the analysis."""
    P = tr.plan
    p = CqtGatherParams(ptr(ws), ptr(tab["kfirst"]), ptr(tab["kcount"]), ptr(tab["rc"]), ptr(tab["Lg"]),
                        ptr(tab["goff"]), ptr(tab["woff"]), ptr(tab["Tk"]), ptr(tab["gdM"] if window is None else window),
                        ptr(X), ptr(cskip), ptr(cout), ptr(hpf), Y.data_ptr(), Y.shape[0], P.Lh, P.ws_per_b, ptr(band_scale))
    call("aid_cqt_gather", p)
