"""Oracle: bit-exact CPU restatement of the product's fp32 convolution arithmetic -- test infrastructure only.

ctypes front end of ``oracle/exact_conv.c`` (built by ``oracle/Makefile`` into ``oracle/_build/libevc_oracle.so``;
``__graft_entry__.build()`` runs that Makefile): every output element is one fixed-order chain of fused multiply-adds,
the order the HIP kernel behind ``EVC_ARITH_F32`` uses.  ``oracle/elic.py`` runs the ELIC entropy-parameter networks
(reference Network.py:132-166) through it when asked for ``exact=True``, so that the oracle and the HIP codec derive
bit-identical means / scales -- and hence identical integer symbols and identical bytes -- from the same stream.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libevc_oracle.so")
_lib = None
_FP = ctypes.POINTER(ctypes.c_float)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):       # a checkout that has not run __graft_entry__.build() yet
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = ctypes.CDLL(_SO)
        _lib.evc_oracle_conv_nhwc_f32.restype = ctypes.c_int
        _lib.evc_oracle_conv_nhwc_f32.argtypes = [_FP, ctypes.c_int, ctypes.c_int, _FP, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, _FP, _FP, _FP, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _FP, ctypes.c_int]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_FP)


def conv_nhwc(x, w, bias=None, res=None, relu_in=False, relu_out=False):
    """x: (B, H, W, Ci) float32 numpy, Ci a multiple of 16; w: (Co, Ci, K, K); "same" zero padding, stride 1."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, H, W, Ci = x.shape
    Co, Ci_w, KH, KW = w.shape
    assert Ci == Ci_w and Ci % 16 == 0, (x.shape, w.shape)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    r = None if res is None else np.ascontiguousarray(res, dtype=np.float32)
    out = np.empty((B, H, W, Co), dtype=np.float32)
    rc = lib().evc_oracle_conv_nhwc_f32(_p(x), Ci, Ci, None, 0, 0, B, H, W, _p(w), _p(b), _p(r), Co, Co, KH, KW,
                                        2 if relu_in else 0, 2 if relu_out else 0, _p(out), Co)
    assert rc == 0
    return out


def _pad_ci(x, w):
    """Zero-pad the channel axis of activations (NHWC) and weights (Co, Ci, K, K) to a multiple of 16, as the product's
    weight packing does (a zero weight times a real activation adds exactly nothing to the chain)."""
    ci = w.shape[1]
    cp = (ci + 15) // 16 * 16
    if cp == ci:
        return x, w
    xp = np.zeros(x.shape[:3] + (cp,), dtype=np.float32)
    xp[..., :ci] = x
    wp = np.zeros((w.shape[0], cp) + w.shape[2:], dtype=np.float32)
    wp[:, :ci] = w
    return xp, wp


def conv2d(x, weight, bias, relu_out=False):
    """torch NCHW in / out; stride 1, padding K // 2."""
    xn = x.permute(0, 2, 3, 1).contiguous().numpy()
    xn, w = _pad_ci(xn, weight.detach().float().numpy())
    out = conv_nhwc(xn, w, bias.detach().float().numpy(), relu_out=relu_out)
    return torch.from_numpy(out).permute(0, 3, 1, 2).contiguous()


def deconv5x5s2(x, weight, bias, relu_out=False):
    """compressai ``deconv`` (ConvTranspose2d k 5, stride 2, padding 2, output_padding 1) in the product's polyphase form
    (evc_amd/csrc/stride2.hip): ONE 3x3 convolution on the low-resolution input with 4 * Cp phase-major output channels,
    bias and activation in its epilogue, then a pure depth-to-space permutation.  weight: (Ci, Co, 5, 5)."""
    wt = weight.detach().float().numpy()
    Ci, Co = wt.shape[:2]
    Cp, CiPad = (Co + 15) // 16 * 16, (Ci + 15) // 16 * 16
    wp = np.zeros((4 * Cp, CiPad, 3, 3), dtype=np.float32)
    for py in (0, 1):
        for px in (0, 1):
            ph = 2 * py + px
            for a in range(3):
                ky = 4 - 2 * a if py == 0 else (-1 if a == 0 else 5 - 2 * a)
                for b in range(3):
                    kx = 4 - 2 * b if px == 0 else (-1 if b == 0 else 5 - 2 * b)
                    if ky >= 0 and kx >= 0:
                        wp[ph * Cp:ph * Cp + Co, :Ci, a, b] = wt[:, :, ky, kx].T
    b4 = np.zeros((4, Cp), dtype=np.float32)
    b4[:, :Co] = bias.detach().float().numpy()[None, :]
    xn = x.permute(0, 2, 3, 1).contiguous().numpy()
    if CiPad != Ci:
        xp = np.zeros(xn.shape[:3] + (CiPad,), dtype=np.float32)
        xp[..., :Ci] = xn
        xn = xp
    y = conv_nhwc(xn, wp, b4.reshape(-1), relu_out=relu_out)             # (B, H, W, 4 Cp)
    B, H, W, _ = y.shape
    out = np.empty((B, 2 * H, 2 * W, Co), dtype=np.float32)
    for py in (0, 1):
        for px in (0, 1):
            ph = 2 * py + px
            out[:, py::2, px::2, :] = y[..., ph * Cp:ph * Cp + Co]
    return torch.from_numpy(out).permute(0, 3, 1, 2).contiguous()
