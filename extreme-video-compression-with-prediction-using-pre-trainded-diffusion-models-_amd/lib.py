"""ctypes bindings to the two C-ABI libraries (include/evc_hip.h, include/evc_rans.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every kernel launch goes
through the C ABI with raw pointers.  There is NO fallback path -- a missing library or a non-gfx950
device raises ``EvcLibraryError`` so a silent eager/CPU substitute can never pass a GPU test.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_longlong, c_uint8, c_void_p

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.path.join(HERE, "lib", "libevc_hip.so")
if os.environ.get("EVC_HIP_SO"):       # A/B builds of the kernel library (tools/ab_lib.sh); in-tree paths only
    HIP_SO = os.path.abspath(os.environ["EVC_HIP_SO"])
RANS_SO = os.path.join(HERE, "lib", "libevc_rans.so")

ACT_NONE, ACT_SILU, ACT_RELU = 0, 1, 2
# How the convolution multiplies (include/evc_hip.h EVC_ARITH_*).  All are fp32 convolutions with fp32 accumulation:
#   F32     v_mfma_f32_32x32x2_f32, exact products;
#   BF16X6  every fp32 operand split exactly into three bf16 values, six cross products on the bf16 matrix cores;
#   F16X3   operands scaled into fp16's range and split 2-way (2^-22 relative), three cross products on the fp16 matrix
#           cores -- only valid where the operand is O(1) (GroupNorm-normalised / activated inputs).
# Measured error vs fp64 on MI355X: F16X3 < BF16X6 < F32 chain (profiles/r02_split_numerics.log).  The packed weights
# carry the choice in their dtype (float32 / bfloat16 / float16).  EVC_CONV_ARITH=f32|bf16x6|f16x3 sets the policy:
# "f16x3" (default) = F16X3 for convolutions whose input is normalised (``bounded_arith``), BF16X6 for the rest.
ARITH_F32, ARITH_BF16X6, ARITH_F16X3 = 0, 1, 2
_ARITH_NAMES = {"f32": ARITH_F32, "bf16x6": ARITH_BF16X6, "f16x3": ARITH_F16X3}
_ARITH_DTYPES = {ARITH_F32: torch.float32, ARITH_BF16X6: torch.bfloat16, ARITH_F16X3: torch.float16}


def _arith_policy():
    name = os.environ.get("EVC_CONV_ARITH", "f16x3").lower()
    if name not in _ARITH_NAMES:
        raise ValueError(f"EVC_CONV_ARITH must be one of {sorted(_ARITH_NAMES)}, got {name!r}")
    return _ARITH_NAMES[name]


def default_arith():
    """Arithmetic for convolutions whose operand range is unknown (raw residual streams, ELIC): never F16X3."""
    a = _arith_policy()
    return ARITH_BF16X6 if a == ARITH_F16X3 else a


def bounded_arith():
    """Arithmetic for convolutions whose input is GroupNorm-normalised / activated, i.e. O(1)."""
    return _arith_policy()


class EvcLibraryError(RuntimeError):
    pass


class EvcKernelError(RuntimeError):
    pass


class ConvArgs(ctypes.Structure):
    """Mirror of ``evc_conv_args`` (include/evc_hip.h)."""
    _fields_ = [("src0", c_void_p), ("src1", c_void_p), ("C0", c_int), ("C1", c_int),
                ("ld0", c_int), ("ld1", c_int),
                ("coef_a", c_void_p), ("coef_s", c_void_p), ("act_in", c_int),
                ("w_packed", c_void_p), ("bias", c_void_p), ("res", c_void_p), ("ld_res", c_int),
                ("out_scale", c_float), ("act_out", c_int),
                ("out", c_void_p), ("ld_out", c_int),
                ("B", c_int), ("H", c_int), ("W", c_int), ("Co", c_int), ("KH", c_int), ("KW", c_int),
                ("splits", c_int), ("stats_out", c_void_p), ("arith", c_int), ("in_bound", c_void_p),
                ("x2_src0", c_void_p), ("x2_src1", c_void_p), ("x2_C0", c_int), ("x2_C1", c_int), ("x2_ld0", c_int),
                ("x2_ld1", c_int), ("x2_w_packed", c_void_p), ("x2_bound", c_void_p)]


# name -> (restype, argtypes); exactly the symbols declared in include/evc_hip.h
HIP_SYMBOLS = {
    "evc_version": (c_char_p, []),
    "evc_arch": (c_char_p, []),
    "evc_device_ok": (c_int, []),
    "evc_clock_probe": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "evc_im2col_nchw_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_void_p]),
    "evc_maxpool3s2_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_lpips_layer_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_upfirdn2d_f32": (c_int, [c_void_p, c_void_p, POINTER(c_float)] + [c_int] * 13 + [c_void_p]),
    "evc_upfirdn2d_nhwc_f32": (c_int, [c_void_p, c_void_p, POINTER(c_float)] + [c_int] * 10 +
                               [c_void_p, c_void_p, c_int, c_void_p]),
    "evc_pack_nchw_to_nhwc_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                          c_void_p]),
    "evc_nhwc_to_nchw_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_chan_stats_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_gn_coeffs_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                  c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    "evc_gn_coeffs_bound_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                        c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "evc_moments_bound_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "evc_attention_f16x3_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_float, c_void_p, c_void_p, c_void_p]),
    "evc_deconv5x5s2_phase_weights_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_conv5x5s2_phase_weights_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "evc_depth_to_space2_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_space_to_depth2_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_deconv5x5s2_workspace_bytes": (c_longlong, [c_int, c_int, c_int, c_int]),
    "evc_deconv5x5s2_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "evc_conv5x5s2_workspace_bytes": (c_longlong, [c_int, c_int, c_int, c_int]),
    "evc_conv5x5s2_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "evc_gdn_workspace_bytes": (c_longlong, [c_int, c_int, c_int, c_int]),
    "evc_gdn_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_void_p]),
    "evc_affine_act_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_void_p]),
    "evc_conv_co_pad": (c_int, [c_int]),
    "evc_conv_packed_floats": (c_longlong, [c_int, c_int, c_int, c_int]),
    "evc_conv_pack_weights_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_conv_packed_bytes": (c_longlong, [c_int, c_int, c_int, c_int, c_int]),
    "evc_conv_pack_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_conv_set_option": (c_int, [c_char_p, c_int]),
    "evc_spade_act_nhwc_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                       c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "evc_conv_choose_splits": (c_int, [POINTER(ConvArgs)]),
    "evc_conv_kernel_name": (c_int, [POINTER(ConvArgs), c_char_p, c_int]),
    "evc_conv_fused_1x1_supported": (c_int, [POINTER(ConvArgs)]),
    "evc_conv_stats_splits": (c_int, [POINTER(ConvArgs)]),
    "evc_conv_workspace_bytes": (c_longlong, [POINTER(ConvArgs)]),
    "evc_conv2d_nhwc_f32": (c_int, [POINTER(ConvArgs), c_void_p, c_void_p]),
    "evc_conv2d_nhwc_profiled_f32": (c_int, [POINTER(ConvArgs), c_void_p, c_void_p, c_void_p, c_void_p]),
    "evc_attention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_float, c_void_p]),
    "evc_attention_workspace_bytes": (c_longlong, [c_int, c_int, c_int, c_int]),
    "evc_attention_set_option": (c_int, [c_char_p, c_int]),
    "evc_attention_ws_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                     c_float, c_void_p, c_void_p]),
    "evc_frame_group_norm_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_float, c_void_p]),
    "evc_frame_attention_f32": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_int] * 5 + [c_float, c_void_p]),
    "evc_frame_mix_f32": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_longlong, c_void_p]),
    "evc_frame_taps_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p]),
    "evc_ddpm_step_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong] + [c_float] * 5 + [c_int, c_void_p]),
    "evc_ddim_step_f32": (c_int, [c_void_p, c_void_p, c_longlong] + [c_float] * 4 + [c_int, c_void_p]),
    "evc_axpy_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_void_p]),
    "evc_pndm_transfer_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_float, c_float, c_float, c_int,
                                      c_void_p]),
    "evc_lincomb4_f32": (c_int, [c_void_p] * 5 + [c_longlong] + [c_float] * 4 + [c_void_p]),
    "evc_scale_clamp_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_float, c_float, c_int, c_float, c_float,
                                    c_void_p]),
    "evc_gate_residual_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p]),
    "evc_elic_gather_params_f32": (c_int, [c_void_p] + [c_int] * 8 + [c_void_p, c_int, c_void_p, c_void_p,
                                                                       c_void_p]),
    "evc_elic_scatter_symbols_f32": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "evc_elic_quantize_f32": (c_int, [c_void_p, c_int, c_int, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p]),
}

RANS_SYMBOLS = {
    "evc_rans_version": (c_char_p, []),
    "evc_rans_max_encoded_bytes": (c_longlong, [c_longlong]),
    "evc_rans_encode_with_indexes": (c_longlong, [POINTER(c_int32), POINTER(c_int32), c_longlong, POINTER(c_int32),
                                                  c_int, POINTER(c_int32), POINTER(c_int32), c_int,
                                                  POINTER(c_uint8), c_longlong]),
    "evc_rans_decode_with_indexes": (c_int, [POINTER(c_uint8), c_longlong, POINTER(c_int32), c_longlong,
                                             POINTER(c_int32), c_int, POINTER(c_int32), POINTER(c_int32), c_int,
                                             POINTER(c_int32)]),
    "evc_pmf_to_quantized_cdf": (c_int, [POINTER(c_float), c_int, c_int, POINTER(c_int32)]),
}

_hip = None
_rans = None


def _load(path, symbols, what):
    if not os.path.exists(path):
        raise EvcLibraryError(f"{what} not built: {path} is missing. Run `python -c \"import __graft_entry__ as g; "
                              f"g.build()\"` (needs hipcc / g++). There is no fallback path.")
    lib = ctypes.CDLL(path)
    for name, (res, args) in symbols.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EvcLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def hip_lib(require_device=True):
    """The HIP kernel library.  ``require_device=False`` is only for symbol/export checks on CPU boxes."""
    global _hip
    if _hip is None:
        _hip = _load(HIP_SO, HIP_SYMBOLS, "libevc_hip.so")
        # A/B switches of the convolution dispatch (same-box comparisons: tools/ab_*.sh): EVC_CONV_OPTIONS="tail_split=0,..."
        for item in filter(None, os.environ.get("EVC_CONV_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            if _hip.evc_conv_set_option(name.strip().encode(), int(value or 1)) != 0:
                raise ValueError(f"EVC_CONV_OPTIONS: unknown convolution option {name!r}")
    if require_device:
        if not torch.cuda.is_available():
            raise EvcLibraryError("no HIP device visible: the evc_amd compute path is gfx950-only (no CPU fallback)")
        if not getattr(hip_lib, "_checked", False):
            if _hip.evc_device_ok() != 1:
                raise EvcLibraryError("current HIP device is not gfx950 (MI355X); libevc_hip.so has no code for it")
            hip_lib._checked = True
    return _hip


def rans_lib():
    global _rans
    if _rans is None:
        _rans = _load(RANS_SO, RANS_SYMBOLS, "libevc_rans.so")
    return _rans


def _check(rc, name):
    if rc != 0:
        raise EvcKernelError(f"{name} failed with code {rc} "
                             "(-1 invalid argument, -2 unsupported, -3 launch failure)")


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-contiguous tensor required"
    return c_void_p(t.data_ptr())


def fptr(t, dtype=torch.float32):
    assert t is None or t.dtype == dtype, (t.dtype, dtype)
    return ptr(t)


# ----------------------------------------------------------------------------------------------
# thin, shape-checked wrappers (one per C entry point that the host code uses)
# ----------------------------------------------------------------------------------------------

def upfirdn2d_nchw(x, kernel, up=1, down=1, pad=(0, 0)):
    """Drop-in for reference ``upfirdn2d(input, kernel, up, down, pad)`` (models/better/op/upfirdn2d.py:13)."""
    L = hip_lib()
    assert x.dim() == 4 and x.dtype == torch.float32
    x = x.contiguous()
    n, c, h, w = x.shape
    k = np.ascontiguousarray(np.asarray(kernel.detach().cpu() if torch.is_tensor(kernel) else kernel,
                                        dtype=np.float32))
    kh, kw = k.shape
    oh = (h * up + pad[0] + pad[1] - kh) // down + 1
    ow = (w * up + pad[0] + pad[1] - kw) // down + 1
    out = torch.empty((n, c, oh, ow), device=x.device, dtype=torch.float32)
    _check(L.evc_upfirdn2d_f32(fptr(x), fptr(out), k.ctypes.data_as(POINTER(c_float)), n * c, h, w, kh, kw, up, up,
                               down, down, pad[0], pad[1], pad[0], pad[1], stream_ptr()), "evc_upfirdn2d_f32")
    return out


def upfirdn2d_nhwc(x, kernel, up, down, pad, coef=None, act=ACT_NONE, out=None):
    L = hip_lib()
    B, H, W, C = x.shape
    k = np.ascontiguousarray(np.asarray(kernel, dtype=np.float32))
    kh, kw = k.shape
    oh = (H * up + pad[0] + pad[1] - kh) // down + 1
    ow = (W * up + pad[0] + pad[1] - kw) // down + 1
    if out is None:
        out = torch.empty((B, oh, ow, C), device=x.device, dtype=torch.float32)
    ca, cs = coef if coef is not None else (None, None)
    _check(L.evc_upfirdn2d_nhwc_f32(fptr(x), fptr(out), k.ctypes.data_as(POINTER(c_float)), B, H, W, C, kh, kw, up,
                                    down, pad[0], pad[1], fptr(ca), fptr(cs), act, stream_ptr()),
           "evc_upfirdn2d_nhwc_f32")
    return out


def pack_nchw_to_nhwc(x0, x1, cpad, out=None):
    L = hip_lib()
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    if out is None:
        out = torch.empty((B, H, W, cpad), device=x0.device, dtype=torch.float32)
    _check(L.evc_pack_nchw_to_nhwc_f32(fptr(x0.contiguous()), C0, fptr(None if x1 is None else x1.contiguous()), C1,
                                       fptr(out), cpad, B, H, W, stream_ptr()), "evc_pack_nchw_to_nhwc_f32")
    return out


def nhwc_to_nchw(x, C, out=None):
    L = hip_lib()
    B, H, W, ld = x.shape
    if out is None:
        out = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    _check(L.evc_nhwc_to_nchw_f32(fptr(x), ld, fptr(out), B, C, H, W, stream_ptr()), "evc_nhwc_to_nchw_f32")
    return out


def stats_splits(B, HW):
    """Number of pixel ranges per image for evc_chan_stats_f32: enough blocks to fill the chip while every
    block still streams >= 64 pixels."""
    n = max(1, min(HW // 64, (1024 + B - 1) // B))
    return n


def chan_stats(x):
    """x: (B, H, W, C) -> partial moments (B, nsplit, C, 2)."""
    L = hip_lib()
    B, H, W, C = x.shape
    ns = stats_splits(B, H * W)
    part = torch.empty((B, ns, C, 2), device=x.device, dtype=torch.float32)
    _check(L.evc_chan_stats_f32(fptr(x), fptr(part), B, H * W, C, ns, stream_ptr()), "evc_chan_stats_f32")
    return part


# Sticky range-event word of the fp16-split arithmetic (include/evc_hip.h EVC_RANGE_*), one per device: the coefficient /
# bound kernels OR bits into it when a tensor holds a NaN / inf or when a GroupNorm-ed operand may leave fp16's range.
RANGE_NONFINITE, RANGE_F16_OPERAND = 1, 2
_range_words = {}


def _events(device):
    w = _range_words.get(device.index)
    if w is None:
        w = _range_words[device.index] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def range_events(device=None, reset=False):
    """Bits raised so far on ``device`` (default: current): 0 = every fp16-split operand was provably in range and
    every tensor finite.  Synchronises the device.  RANGE_F16_OPERAND set means a trained checkpoint drives a normalised
    activation towards fp16's limit: run that model with EVC_CONV_ARITH=bf16x6 (exact 3-way bf16 split, no range limit)."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    w = _range_words.get(idx)
    if w is None:
        return 0
    v = int(w.item())
    if reset:
        w.zero_()
    return v


def im2col_nchw(x, KH, KW, stride, pad, ld_out, shift=None, scale=None):
    """x: (N, C, H, W) -> (N, Ho, Wo, ld_out) patch rows in (c, ky, kx) order, zero-filled beyond C*KH*KW; optional
    (x - shift[c]) / scale[c] before the zero padding (include/evc_hip.h)."""
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    out = torch.empty((N, Ho, Wo, ld_out), device=x.device, dtype=torch.float32)
    _check(hip_lib().evc_im2col_nchw_f32(fptr(x), fptr(out), N, C, H, W, KH, KW, stride, pad, ld_out, ptr(shift), ptr(scale),
                                         stream_ptr()), "evc_im2col_nchw_f32")
    return out


def maxpool3s2_nhwc(x):
    N, H, W, C = x.shape
    out = torch.empty((N, (H - 3) // 2 + 1, (W - 3) // 2 + 1, C), device=x.device, dtype=torch.float32)
    _check(hip_lib().evc_maxpool3s2_nhwc_f32(fptr(x), fptr(out), N, H, W, C, stream_ptr()), "evc_maxpool3s2_nhwc_f32")
    return out


def lpips_layer(f0, f1, lin_w, dist, accumulate):
    """dist[n] (+)= spatial mean of lin_w . (unit-normalised f0 - unit-normalised f1)^2; f0, f1: (N, H, W, C)."""
    N, H, W, C = f0.shape
    assert f1.shape == f0.shape and lin_w.numel() == C and dist.numel() == N
    _check(hip_lib().evc_lpips_layer_f32(fptr(f0), fptr(f1), fptr(lin_w), fptr(dist), N, H * W, C, int(bool(accumulate)),
                                         stream_ptr()), "evc_lpips_layer_f32")
    return dist


class ClockProbe:
    """Shader clock held by the chip while other streams work, measured by a one-wave idle kernel on its own stream
    (``evc_clock_probe``): start it, enqueue the work to be characterised, call ``stop()`` (a stream-ordered write behind that
    work on the CURRENT stream: the probe ends with it, or after ``max_us`` at the latest), then read ``ghz()``."""

    def __init__(self, max_us, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.out = torch.zeros(2, dtype=torch.int64, device=self.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))          # the two zero-fills above
        _check(hip_lib().evc_clock_probe(self.out.data_ptr(), int(max_us), self.flag.data_ptr(), self.stream.cuda_stream),
               "evc_clock_probe")

    def stop(self):
        self.flag.fill_(1)

    def ghz(self):
        self.stream.synchronize()
        ticks, ref = (int(v) for v in self.out.tolist())
        return ticks / ref * 0.1 if ref else None

    def seconds(self):
        return int(self.out[1]) / 1e8


def gpu_power_w(device=None):
    """Average package power (W) of ``device`` from sysfs (hwmon power1_average of the card with the device's PCI address), or
    None where that is not readable.  Read-only; the figure is the driver's ~1 s moving average."""
    import glob as _glob
    try:
        pr = torch.cuda.get_device_properties(torch.cuda.current_device() if device is None else torch.device(device).index)
        addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        for card in _glob.glob("/sys/class/drm/card*/device"):
            if addr in os.path.realpath(card):
                for f in _glob.glob(card + "/hwmon/hwmon*/power1_average") + _glob.glob(card + "/hwmon/hwmon*/power1_input"):
                    return int(open(f).read()) / 1e6
    except Exception:
        return None
    return None


def gn_coeffs(parts, HW, groups, eps, mode=0, gamma=None, beta=None, ss=None, row=None, bound=None):
    """parts: one or two partial-moment tensors (virtual concat). Returns (coef_a, coef_s), each (B, C).
    ``bound``: optional one-element int32 tensor (zeroed by the caller) raised to the bit pattern of the tensors'
    element bound S (|x| <= sqrt(S)) -- what ``conv2d_nhwc(..., in_bound=)`` takes."""
    L = hip_lib()
    p0 = parts[0]
    p1 = parts[1] if len(parts) > 1 else None
    B, ns0, C0, _ = p0.shape
    ns1, C1 = (p1.shape[1], p1.shape[2]) if p1 is not None else (0, 0)
    C = C0 + C1
    ca = torch.empty((B, C), device=p0.device, dtype=torch.float32)
    cs = torch.empty((B, C), device=p0.device, dtype=torch.float32)
    ss_ld = 0 if ss is None else ss.stride(0)
    _check(L.evc_gn_coeffs_bound_f32(fptr(p0), ns0, C0, fptr(p1), ns1, C1, B, HW, groups, eps, mode, fptr(gamma),
                                     fptr(beta), c_void_p(ss.data_ptr()) if ss is not None else None, ss_ld,
                                     fptr(row, torch.int32), fptr(ca), fptr(cs), _word(bound), _word(_events(p0.device)),
                                     stream_ptr()),
           "evc_gn_coeffs_bound_f32")
    return ca, cs


def _word(t, n=1):
    """Device pointer of an n-element int32 tensor (or view), None when absent."""
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.int32 and t.numel() == n and t.is_contiguous()
    return c_void_p(t.data_ptr())


def moments_bound(part, c_begin, c_count, bound):
    """Raise ``bound[z]`` to the element bound of channels [c_begin + z*c_count, + c_count) of a moments tensor
    (B, ns, C, 2), for z < bound.numel()."""
    B, ns, C, _ = part.shape
    n = bound.numel()
    _check(hip_lib().evc_moments_bound_f32(fptr(part), ns, C, c_begin, c_count, n, B, _word(bound, n),
                                           _word(_events(part.device)), stream_ptr()),
           "evc_moments_bound_f32")


def affine_act(x, coef, act, out=None, coef_col=0):
    """y = act(x * a + s).  ``out`` may be a ``Cols`` slice of a wider buffer; ``coef_col`` selects the channel
    offset inside wider (B, Ctot) coefficient tensors (both let a concat be activated piecewise)."""
    L = hip_lib()
    B, H, W, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    po, Cout, ldo, _ = _src(out)
    assert Cout == C
    ca, cs = coef if coef is not None else (None, None)
    ldc = 0 if ca is None else ca.shape[-1]
    pa = None if ca is None else c_void_p(ca.data_ptr() + 4 * coef_col)
    ps = None if cs is None else c_void_p(cs.data_ptr() + 4 * coef_col)
    _check(L.evc_affine_act_nhwc_f32(fptr(x), po, pa, ps, act, B, H * W, C, ldc, ldo, stream_ptr()),
           "evc_affine_act_nhwc_f32")
    return out.t if isinstance(out, Cols) else out


def spade_act(x, coef, maps, ctot, col=0, ss=None, row=None, act=ACT_SILU, out=None):
    """SPADE act-norm of one part of a (virtual) concat: ``x`` (B, H, W, C) contiguous; ``coef`` = (a, s), each (B, ctot);
    ``maps`` (B, H, W, 2*ctot) = [1 + gamma | beta]; ``ss`` the AdaGN table slice (rows, 2*ctot) = [scale | shift] (None:
    no time embedding); ``col`` the part's channel offset inside the concat; ``out`` a tensor or ``Cols`` slice."""
    L = hip_lib()
    B, H, W, C = x.shape
    assert x.is_contiguous() and maps.is_contiguous() and maps.shape == (B, H, W, 2 * ctot) and col + C <= ctot
    if out is None:
        out = torch.empty_like(x)
    po, Cout, ldo, _ = _src(out)
    assert Cout == C
    ca, cs = coef
    assert ca.shape == (B, ctot) and cs.shape == (B, ctot)
    off = lambda t, c: c_void_p(t.data_ptr() + 4 * c)
    ps = psh = None
    ld_ss = 0
    if ss is not None:
        assert ss.shape[1] == 2 * ctot and ss.stride(1) == 1
        ps, psh, ld_ss = off(ss, col), off(ss, ctot + col), ss.stride(0)
    _check(L.evc_spade_act_nhwc_f32(fptr(x), po, off(ca, col), off(cs, col), ctot, off(maps, col), off(maps, ctot + col),
                                    2 * ctot, ps, psh, ld_ss, fptr(row, torch.int32), act, B, H * W, C, ldo, stream_ptr()),
           "evc_spade_act_nhwc_f32")
    return out.t if isinstance(out, Cols) else out


def conv_pack_weights(w, arith=None):
    """w: (Co, Ci, KH, KW) device float32 with Ci % 16 == 0 -> packed flat tensor: float32 for ARITH_F32, bfloat16
    (three planes) for ARITH_BF16X6, float16 (header + two planes) for ARITH_F16X3; ``conv2d_nhwc`` picks the kernel
    from that dtype."""
    L = hip_lib()
    arith = default_arith() if arith is None else arith
    w = w.contiguous()
    Co, Ci, KH, KW = w.shape
    nbytes = L.evc_conv_packed_bytes(Co, Ci, KH, KW, arith)
    if nbytes < 0:
        raise EvcKernelError(f"evc_conv_packed_bytes rejected the arguments ({nbytes})")
    if arith not in _ARITH_DTYPES:
        raise EvcKernelError(f"unknown convolution arithmetic {arith}")
    dt = _ARITH_DTYPES[arith]
    packed = torch.empty((nbytes // dt.itemsize,), device=w.device, dtype=dt)
    _check(L.evc_conv_pack_weights(fptr(w), ptr(packed), Co, Ci, KH, KW, arith, stream_ptr()), "evc_conv_pack_weights")
    return packed


def conv_set_option(name, value):
    """Dispatch switches of the convolution ("wide_tiles", "row_reuse", "tail_split", "wide256"): include/evc_hip.h."""
    _check(hip_lib(require_device=False).evc_conv_set_option(name.encode(), int(value)), "evc_conv_set_option")


def packed_arith(w_packed):
    return {torch.bfloat16: ARITH_BF16X6, torch.float16: ARITH_F16X3}.get(w_packed.dtype, ARITH_F32)


_ws_cache = {}

# Optional launch profiler (bench.py roofline leg): when a list is installed here every conv launch is
# bracketed by HIP events recorded on the stream the kernel runs on.
CONV_PROFILE = None


def conv_variant(Co):
    """Which conv_igemm_kernel<TN> instantiation serves an output width (mirrors pick_tn in conv_igemm.hip)."""
    cp = (Co + 63) // 64 * 64
    return 3 if cp % 192 == 0 else (2 if cp % 128 == 0 else 1)


def _workspace(nbytes, device):
    """Split-K workspace.  Eager launches: one grow-only buffer per (device, stream), reused stream-ordered.
    Inside a HIP-graph capture every graph must own its buffer (all captures share one capture stream, and
    graphs replayed concurrently on different streams would otherwise race on it): allocate from the graph pool."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(((nbytes + 3) // 4,), device=device, dtype=torch.float32)
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    cur = _ws_cache.get(key)
    if cur is None or cur.numel() * 4 < nbytes:
        cur = torch.empty(((nbytes + 3) // 4,), device=device, dtype=torch.float32)
        _ws_cache[key] = cur
    return cur


class Cols:
    """A channel slice [c0, c0 + C) of a contiguous NHWC tensor, usable as a conv source or destination."""
    __slots__ = ("t", "c0", "C")

    def __init__(self, t, c0, C):
        assert t.is_contiguous() and 0 <= c0 and c0 + C <= t.shape[-1] and c0 % 4 == 0
        self.t, self.c0, self.C = t, c0, C


def _src(s):
    """-> (pointer, channels, row stride, shape[:3])"""
    if s is None:
        return None, 0, 0, None
    if isinstance(s, Cols):
        return c_void_p(s.t.data_ptr() + 4 * s.c0), s.C, s.t.shape[-1], s.t.shape[:3]
    return ptr(s), s.shape[-1], s.shape[-1], s.shape[:3]


def conv_fused_1x1_supported(B, H, W, Ci, Co, arith, splits=0):
    """Whether a 3x3 convolution of this shape may carry a fused 1x1 operand (``conv2d_nhwc(..., x2=)``): C query.  It
    needs the f16x3 row-reuse kernel, i.e. 128-pixel tiles: very small grids, for which 64-pixel tiles are chosen, do not."""
    d = c_void_p(16)
    a = ConvArgs(d, None, Ci, 0, 0, 0, None, None, ACT_NONE, d, None, None, 0, 1.0, ACT_NONE, d, Co, B, H, W, Co, 3, 3, splits,
                 None, arith, None)
    return bool(hip_lib(require_device=False).evc_conv_fused_1x1_supported(ctypes.byref(a)))


def conv2d_nhwc(src0, w_packed, Co, KH, KW, bias=None, src1=None, coef=None, act_in=ACT_NONE, res=None,
                out_scale=1.0, act_out=ACT_NONE, out=None, splits=0, want_stats=False, in_bound=None, x2=None):
    """out = act_out((conv(act_in(cat[src0,src1]*a+s), w) + bias + res) * out_scale); tensors are NHWC.
    ``src0`` / ``src1`` / ``out`` may be ``Cols`` channel slices of wider tensors.
    ``x2 = (x2_src0, x2_src1 or None, w2_packed, bound)``: a fused 1x1 operand -- conv1x1(cat[x2_src0, x2_src1], w2) is
    accumulated into the same output (include/evc_hip.h); ``bias`` must then be the sum of both convolutions' biases.
    ``want_stats=True`` returns ``(out, stats)``: per-channel moments of ``out`` in ``chan_stats`` layout, produced
    by the conv epilogue when the shape allows it, else by a separate ``evc_chan_stats_f32`` pass."""
    L = hip_lib()
    p0, C0, ld0, shp = _src(src0)
    p1, C1, ld1, shp1 = _src(src1)
    assert shp1 is None or tuple(shp1) == tuple(shp)
    B, H, W = shp
    dev = (src0.t if isinstance(src0, Cols) else src0).device
    if out is None:
        out = torch.empty((B, H, W, Co), device=dev, dtype=torch.float32)
    po, Cout, ldo, shpo = _src(out)
    assert tuple(shpo) == (B, H, W) and Cout >= Co
    ca, cs = coef if coef is not None else (None, None)
    a = ConvArgs(p0, p1, C0, C1, ld0, ld1, ptr(ca), ptr(cs), act_in, ptr(w_packed), ptr(bias), ptr(res),
                 0 if res is None else res.shape[-1], float(out_scale), act_out, po, ldo,
                 B, H, W, Co, KH, KW, splits, None, packed_arith(w_packed), _word(in_bound))
    x2_ci = 0
    if x2 is not None:
        q0, qC0, qld0, qshp = _src(x2[0])
        q1, qC1, qld1, _ = _src(x2[1])
        assert tuple(qshp) == (B, H, W) and x2[2].dtype == torch.float16
        a.x2_src0, a.x2_src1, a.x2_C0, a.x2_C1, a.x2_ld0, a.x2_ld1 = q0, q1, qC0, qC1, qld0, qld1
        a.x2_w_packed, a.x2_bound = ptr(x2[2]), _word(x2[3])
        x2_ci = qC0 + qC1
    stats = None
    if want_stats:
        ns = L.evc_conv_stats_splits(ctypes.byref(a))
        if ns > 0:
            stats = torch.empty((B, ns, Co, 2), device=dev, dtype=torch.float32)
            a.stats_out = ptr(stats)
    nbytes = L.evc_conv_workspace_bytes(ctypes.byref(a))
    if nbytes < 0:
        raise EvcKernelError(f"evc_conv_workspace_bytes rejected the arguments ({nbytes})")
    ws = _workspace(nbytes, dev) if nbytes > 0 else None
    if CONV_PROFILE is not None:
        # e0 .. ec: the convolution kernel alone (recorded inside the C call, before the split-K combine); e0 .. e1: with it
        st = torch.cuda.current_stream()
        e0, ec, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(st); ec.record(st)          # (a torch event owns a hipEvent_t only once it has been recorded)
        _check(L.evc_conv2d_nhwc_profiled_f32(ctypes.byref(a), ptr(ws), stream_ptr(), c_void_p(e0.cuda_event),
                                              c_void_p(ec.cuda_event)), "evc_conv2d_nhwc_profiled_f32")
        e1.record(st)
        kbuf = ctypes.create_string_buffer(96)
        L.evc_conv_kernel_name(ctypes.byref(a), kbuf, 96)
        CONV_PROFILE.append(dict(variant=conv_variant(Co), split=nbytes > 0, e0=e0, ec=ec, e1=e1, arith=a.arith,
                                 kernel=kbuf.value.decode(),
                                 flops=2.0 * B * H * W * Co * (KH * KW * (C0 + C1) + x2_ci),
                                 shape=(B, H, W, C0 + C1, Co, KH),
                                 call=dict(B=B, H=H, W=W, C0=C0, C1=C1, Co=Co, K=KH, coef=coef is not None, act_in=act_in,
                                           res=res is not None, out_scale=float(out_scale), bias=bias is not None,
                                           bound=in_bound is not None, arith=a.arith, ld_out=ldo,
                                           stats=bool(want_stats), x2=x2_ci)))
    else:
        _check(L.evc_conv2d_nhwc_f32(ctypes.byref(a), ptr(ws), stream_ptr()), "evc_conv2d_nhwc_f32")
    result = out.t if isinstance(out, Cols) else out
    if want_stats:
        return result, (stats if stats is not None else chan_stats(result))
    return result


def conv_fused_stats_splits(B, H, W, Ci, Co, KH, KW, splits=0, arith=None):
    """HW/64 or HW/32 when ``conv2d_nhwc(..., want_stats=True)`` gets its moments from the conv epilogue / split-K
    combine for this shape, 0 when it falls back to a separate ``evc_chan_stats_f32`` pass (C query, no launch)."""
    d = c_void_p(16)   # any non-null pointers: the query validates shapes only
    a = ConvArgs(d, None, Ci, 0, 0, 0, None, None, ACT_NONE, d, None, None, 0, 1.0, ACT_NONE, d, Co, B, H, W, Co, KH,
                 KW, splits, None, default_arith() if arith is None else arith, None)
    return hip_lib(require_device=False).evc_conv_stats_splits(ctypes.byref(a))


def conv_workspace_bytes(B, H, W, Ci, Co, KH, KW, splits=0, arith=None):
    """Bytes of split-K slabs ``conv2d_nhwc`` takes for this shape: 0 for an unsplit grid, splits x output for split-K,
    tail splits x tail rows when only the last partial round of tiles is split (C query, no launch)."""
    d = c_void_p(16)
    a = ConvArgs(d, None, Ci, 0, 0, 0, None, None, ACT_NONE, d, None, None, 0, 1.0, ACT_NONE, d, Co, B, H, W, Co, KH,
                 KW, splits, None, default_arith() if arith is None else arith, None)
    return hip_lib(require_device=False).evc_conv_workspace_bytes(ctypes.byref(a))


def attention_set_option(name, value):
    """A/B switch of the attention kernels (include/evc_hip.h: "kv_planes")."""
    if hip_lib().evc_attention_set_option(name.encode(), int(value)) != 0:
        raise EvcKernelError(f"unknown attention option {name!r}")


def attention(qkv, C, heads, out=None, bounds=None):
    """qkv: (B, N, 3C) with q | k | v concatenated along channels; returns (B, N, C).  ``bounds``: three int32 words
    (element bounds of q, k, v from ``moments_bound``) select the fp16-split kernel; None the f32-MFMA one."""
    L = hip_lib()
    B, N, ld = qkv.shape
    D = C // heads
    if out is None:
        out = torch.empty((B, N, C), device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    nbytes = L.evc_attention_workspace_bytes(B, heads, N, D)
    ws = _workspace(nbytes, qkv.device) if nbytes > 0 else None      # stream-ordered: shared with the conv workspace
    if bounds is not None:
        _check(L.evc_attention_f16x3_f32(c_void_p(base), c_void_p(base + 4 * C), c_void_p(base + 8 * C), ld, fptr(out), C,
                                         B, heads, N, D, float(int(D) ** (-0.5)), _word(bounds, 3), ptr(ws), stream_ptr()),
               "evc_attention_f16x3_f32")
        return out
    _check(L.evc_attention_ws_f32(c_void_p(base), c_void_p(base + 4 * C), c_void_p(base + 8 * C), ld, fptr(out), C, B,
                                  heads, N, D, float(int(D) ** (-0.5)), ptr(ws), stream_ptr()), "evc_attention_ws_f32")
    return out


def frame_group_norm(x, N, gamma, beta, groups, eps):
    """GroupNorm over (C / groups channels x N frames) of each pixel, affine.  x: (B*N, H, W, C), a sample's frames adjacent."""
    BN, H, W, C = x.shape
    assert BN % N == 0 and x.is_contiguous()
    y = torch.empty_like(x)
    _check(hip_lib().evc_frame_group_norm_f32(fptr(x), fptr(y), fptr(gamma), fptr(beta), BN // N, N, H * W, C, groups,
                                              float(eps), stream_ptr()), "evc_frame_group_norm_f32")
    return y


def frame_attention(qkv, N, C, heads):
    """Attention over the N frames of each pixel.  qkv: (B*N, H, W, 3C) with q | k | v along channels; returns (B*N, H, W, C)."""
    BN, H, W, ld = qkv.shape
    assert BN % N == 0 and ld == 3 * C and qkv.is_contiguous()
    out = torch.empty((BN, H, W, C), device=qkv.device, dtype=torch.float32)
    _check(hip_lib().evc_frame_attention_f32(fptr(qkv), ld, fptr(out), C, BN // N, N, H * W, C, heads,
                                             float(int(C // heads) ** (-0.5)), stream_ptr()), "evc_frame_attention_f32")
    return out


def frame_mix(x, N, w, bias):
    """1x1 convolution over the frame axis: x (B*N, H, W, C), w (M, N) -> (B*M, H, W, C)."""
    BN, H, W, C = x.shape
    M = w.shape[0]
    assert BN % N == 0 and tuple(w.shape) == (M, N) and x.is_contiguous() and w.is_contiguous()
    B = BN // N
    y = torch.empty((B * M, H, W, C), device=x.device, dtype=torch.float32)
    _check(hip_lib().evc_frame_mix_f32(fptr(x), fptr(y), fptr(w), fptr(bias), B, N, M, H * W * C, stream_ptr()),
           "evc_frame_mix_f32")
    return y


def frame_taps(x, N):
    """Frames n - 1 | n | n + 1 side by side along the channels (zeros beyond a sample's frames): (B*N, H, W, C) -> (B*N, H, W, 3C)."""
    BN, H, W, C = x.shape
    assert BN % N == 0 and x.is_contiguous()
    y = torch.empty((BN, H, W, 3 * C), device=x.device, dtype=torch.float32)
    _check(hip_lib().evc_frame_taps_f32(fptr(x), fptr(y), BN // N, N, H * W, C, stream_ptr()), "evc_frame_taps_f32")
    return y


class Deconv5x5s2:
    """compressai ``deconv`` (ConvTranspose2d k 5, stride 2, padding 2, output_padding 1) in polyphase form
    (include/evc_hip.h, csrc/stride2.hip).  ``weight``: the module's (Ci, Co, 5, 5) tensor, ``bias``: (Co,)."""

    def __init__(self, weight, bias, arith=None, device="cuda"):
        Lh = hip_lib()
        w = weight.detach().to(device, torch.float32).contiguous()
        self.Ci, self.Co = w.shape[0], w.shape[1]
        self.Cp, self.CiPad = (self.Co + 15) // 16 * 16, (self.Ci + 15) // 16 * 16
        wp = torch.empty((4 * self.Cp, self.CiPad, 3, 3), device=device, dtype=torch.float32)
        _check(Lh.evc_deconv5x5s2_phase_weights_f32(fptr(w), fptr(wp), self.Ci, self.Co, self.Cp, self.CiPad, stream_ptr()),
               "evc_deconv5x5s2_phase_weights_f32")
        self.arith = default_arith() if arith is None else arith
        self.w = conv_pack_weights(wp, self.arith)
        b4 = torch.zeros((4, self.Cp), device=device, dtype=torch.float32)
        b4[:, :self.Co] = bias.detach().to(device, torch.float32)[None, :]
        self.bias4 = b4.reshape(-1).contiguous()

    def __call__(self, x, act_out=ACT_NONE):
        B, H, W, C = x.shape
        assert C == self.CiPad and x.dtype == torch.float32 and x.is_contiguous()
        Lh = hip_lib()
        ws = _workspace(Lh.evc_deconv5x5s2_workspace_bytes(B, H, W, self.Cp), x.device)
        out = torch.empty((B, 2 * H, 2 * W, self.Co), device=x.device, dtype=torch.float32)
        _check(Lh.evc_deconv5x5s2_f32(fptr(x), ptr(self.w), self.arith, fptr(self.bias4), fptr(out), ptr(ws), B, H, W, C,
                                      self.Co, act_out, stream_ptr()), "evc_deconv5x5s2_f32")
        return out


class Conv5x5s2:
    """compressai ``conv`` (Conv2d k 5, stride 2, padding 2) in polyphase form.  ``weight``: (Co, Ci, 5, 5)."""

    def __init__(self, weight, bias, arith=None, device="cuda"):
        Lh = hip_lib()
        w = weight.detach().to(device, torch.float32).contiguous()
        self.Co, self.Ci = w.shape[0], w.shape[1]
        self.Cq = (self.Ci + 3) // 4 * 4
        wq = torch.empty((self.Co, 4 * self.Cq, 3, 3), device=device, dtype=torch.float32)
        _check(Lh.evc_conv5x5s2_phase_weights_f32(fptr(w), fptr(wq), self.Co, self.Ci, self.Cq, stream_ptr()),
               "evc_conv5x5s2_phase_weights_f32")
        self.arith = default_arith() if arith is None else arith
        self.w = conv_pack_weights(wq, self.arith)
        self.bias = bias.detach().to(device, torch.float32).contiguous()

    def __call__(self, x, act_out=ACT_NONE, channels=None):
        """x: (B, 2Ho, 2Wo, ld) of which the first ``channels`` (default all) are the input."""
        B, H2, W2, ld = x.shape
        C = ld if channels is None else channels
        assert C == self.Ci and ld >= C and H2 % 2 == 0 and W2 % 2 == 0 and x.is_contiguous()
        Lh = hip_lib()
        ws = _workspace(Lh.evc_conv5x5s2_workspace_bytes(B, H2 // 2, W2 // 2, C), x.device)
        out = torch.empty((B, H2 // 2, W2 // 2, self.Co), device=x.device, dtype=torch.float32)
        _check(Lh.evc_conv5x5s2_f32(fptr(x), ld, ptr(self.w), self.arith, fptr(self.bias), fptr(out), ptr(ws), B, H2 // 2,
                                    W2 // 2, C, self.Co, act_out, stream_ptr()), "evc_conv5x5s2_f32")
        return out


class GDN:
    """GDN / IGDN / GDN1 (reference ELICUtilis/layers/gdn.py:26-106) on NHWC tensors.  ``beta`` (C,) and ``gamma`` (C, C)
    are the RAW parameters of the reference module; compressai's NonNegativeParametrizer is applied here, once:
    ``max(p, sqrt(minimum + 2^-36))^2 - 2^-36`` (minimum = beta_min for beta, 0 for gamma)."""
    PEDESTAL = (2.0 ** -18) ** 2

    def __init__(self, beta, gamma, inverse=False, simplified=False, beta_min=1e-6, device="cuda"):
        hip_lib()
        b = beta.detach().to(device, torch.float32)
        g = gamma.detach().to(device, torch.float32)
        self.C = b.numel()
        self.beta = (torch.clamp(b, min=(beta_min + self.PEDESTAL) ** 0.5) ** 2 - self.PEDESTAL).contiguous()
        gm = torch.clamp(g, min=self.PEDESTAL ** 0.5) ** 2 - self.PEDESTAL
        self.arith = default_arith()
        self.gamma = conv_pack_weights(gm.reshape(self.C, self.C, 1, 1).contiguous(), self.arith)
        self.inverse, self.simplified = bool(inverse), bool(simplified)

    def __call__(self, x, out=None):
        B, H, W, C = x.shape
        assert C == self.C and x.dtype == torch.float32
        Lh = hip_lib()
        nbytes = Lh.evc_gdn_workspace_bytes(B, H, W, C)
        if nbytes < 0:
            raise EvcKernelError(f"evc_gdn_workspace_bytes rejected the arguments ({nbytes})")
        ws = _workspace(nbytes, x.device)
        if out is None:
            out = torch.empty_like(x)
        _check(Lh.evc_gdn_f32(fptr(x), ptr(self.gamma), self.arith, fptr(self.beta), fptr(out), ptr(ws), B, H, W, C,
                              int(self.inverse), int(self.simplified), stream_ptr()), "evc_gdn_f32")
        return out


def ddpm_step(x, e, noise, k1, k2, c1, c2, sigma, clip):
    _check(hip_lib().evc_ddpm_step_f32(fptr(x), fptr(e), fptr(noise), x.numel(), k1, k2, c1, c2, sigma, int(clip),
                                       stream_ptr()), "evc_ddpm_step_f32")


def ddim_step(x, e, k1, k2, c1, c2, clip):
    _check(hip_lib().evc_ddim_step_f32(fptr(x), fptr(e), x.numel(), k1, k2, c1, c2, int(clip), stream_ptr()),
           "evc_ddim_step_f32")


def axpy(x, e, alpha, out=None):
    if out is None:
        out = torch.empty_like(x)
    _check(hip_lib().evc_axpy_f32(fptr(x), fptr(e), fptr(out), x.numel(), alpha, stream_ptr()), "evc_axpy_f32")
    return out


def pndm_transfer(x, e, d, cx, ce, clip, out=None):
    if out is None:
        out = torch.empty_like(x)
    _check(hip_lib().evc_pndm_transfer_f32(fptr(x), fptr(e), fptr(out), x.numel(), d, cx, ce, int(clip),
                                           stream_ptr()), "evc_pndm_transfer_f32")
    return out


def lincomb4(es, ws, out=None):
    es = list(es) + [None] * (4 - len(es))
    ws = list(ws) + [0.0] * (4 - len(ws))
    if out is None:
        out = torch.empty_like(es[0])
    _check(hip_lib().evc_lincomb4_f32(fptr(es[0]), fptr(es[1]), fptr(es[2]), fptr(es[3]), fptr(out), out.numel(),
                                      *[float(w) for w in ws], stream_ptr()), "evc_lincomb4_f32")
    return out


def scale_clamp(x, mul, add, clamp=None, out=None):
    if out is None:
        out = torch.empty_like(x)
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    _check(hip_lib().evc_scale_clamp_f32(fptr(x), fptr(out), x.numel(), mul, add, int(clamp is not None), lo, hi,
                                         stream_ptr()), "evc_scale_clamp_f32")
    return out


def gate_residual(a, b, x, out=None):
    if out is None:
        out = torch.empty_like(x)
    _check(hip_lib().evc_gate_residual_f32(fptr(a), fptr(b), fptr(x), fptr(out), x.numel(), stream_ptr()),
           "evc_gate_residual_f32")
    return out


def elic_gather_params(ms, mean_off, scale_off, C, parity, scale_table):
    """ms: (B, H, W, ld) -> (idx int32 (B, C, H, W/2), means float32 (B, C, H, W/2))."""
    B, H, W, ld = ms.shape
    idx = torch.empty((B, C, H, W // 2), device=ms.device, dtype=torch.int32)
    means = torch.empty((B, C, H, W // 2), device=ms.device, dtype=torch.float32)
    _check(hip_lib().evc_elic_gather_params_f32(fptr(ms), ld, mean_off, scale_off, C, B, H, W, parity,
                                                fptr(scale_table), scale_table.numel(), fptr(idx, torch.int32),
                                                fptr(means), stream_ptr()), "evc_elic_gather_params_f32")
    return idx, means


def elic_scatter_symbols(symbols, means, y_hat, c0, parity):
    B, C, H, Wh = symbols.shape
    _check(hip_lib().evc_elic_scatter_symbols_f32(fptr(symbols, torch.int32), fptr(means), fptr(y_hat),
                                                  y_hat.shape[-1], c0, C, B, H, 2 * Wh, parity, stream_ptr()),
           "evc_elic_scatter_symbols_f32")


def elic_quantize(y, c0, means, parity):
    """y: (B, H, W, ld); means: (B, C, H, W/2) -> int32 symbols (B, C, H, W/2)."""
    B, C, H, Wh = means.shape
    sym = torch.empty((B, C, H, Wh), device=y.device, dtype=torch.int32)
    _check(hip_lib().evc_elic_quantize_f32(fptr(y), y.shape[-1], c0, fptr(means), C, B, H, 2 * Wh, parity,
                                           fptr(sym, torch.int32), stream_ptr()), "evc_elic_quantize_f32")
    return sym


# ---- host rANS -------------------------------------------------------------------------------

def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def rans_encode(symbols, indexes, cdfs, cdf_sizes, offsets):
    """-> bytes.  Arguments are int arrays; ``cdfs`` is (n_cdfs, ld)."""
    L = rans_lib()
    s, i, c, z, o = _i32(symbols).ravel(), _i32(indexes).ravel(), _i32(cdfs), _i32(cdf_sizes).ravel(), _i32(offsets).ravel()
    cap = L.evc_rans_max_encoded_bytes(s.size)
    out = np.empty(cap, dtype=np.uint8)
    P32, P8 = POINTER(c_int32), POINTER(c_uint8)
    n = L.evc_rans_encode_with_indexes(s.ctypes.data_as(P32), i.ctypes.data_as(P32), s.size, c.ctypes.data_as(P32),
                                       c.shape[1], z.ctypes.data_as(P32), o.ctypes.data_as(P32), c.shape[0],
                                       out.ctypes.data_as(P8), cap)
    if n < 0:
        raise EvcKernelError(f"evc_rans_encode_with_indexes failed ({n})")
    return out[:n].tobytes()


def rans_decode(data, indexes, cdfs, cdf_sizes, offsets):
    """-> int32 array of len(indexes) symbols."""
    L = rans_lib()
    buf = np.frombuffer(data, dtype=np.uint8)
    i, c, z, o = _i32(indexes).ravel(), _i32(cdfs), _i32(cdf_sizes).ravel(), _i32(offsets).ravel()
    out = np.empty(i.size, dtype=np.int32)
    P32, P8 = POINTER(c_int32), POINTER(c_uint8)
    rc = L.evc_rans_decode_with_indexes(buf.ctypes.data_as(P8), buf.size, i.ctypes.data_as(P32), i.size,
                                        c.ctypes.data_as(P32), c.shape[1], z.ctypes.data_as(P32),
                                        o.ctypes.data_as(P32), c.shape[0], out.ctypes.data_as(P32))
    if rc != 0:
        raise EvcKernelError(f"evc_rans_decode_with_indexes failed ({rc})")
    return out


def pmf_to_quantized_cdf(pmf, precision=16):
    L = rans_lib()
    p = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.empty(p.size + 1, dtype=np.int32)
    rc = L.evc_pmf_to_quantized_cdf(p.ctypes.data_as(POINTER(c_float)), p.size, precision,
                                    out.ctypes.data_as(POINTER(c_int32)))
    if rc != 0:
        raise EvcKernelError(f"evc_pmf_to_quantized_cdf failed ({rc})")
    return out
