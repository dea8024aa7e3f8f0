"""Oracle: upfirdn2d (upsample -> pad -> FIR -> downsample), test infrastructure only.

Numpy restatement of reference ``models/better/op/upfirdn2d.py:163-204`` (upfirdn2d_native) and the
two call shapes of ``models/better/up_or_down_sampling.py:182-258`` (upsample_2d / downsample_2d).
Loops over taps explicitly so it shares nothing with the conv-based reference formulation.
"""
import numpy as np


def setup_kernel(k):
    """up_or_down_sampling.py:182-189."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k /= np.sum(k)
    return k


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """x: (N, C, H, W) float32, kernel: (kh, kw).  Same maths as upfirdn2d_native with
    up_x=up_y=up, down_x=down_y=down, pad_x0=pad_y0=pad[0], pad_x1=pad_y1=pad[1]."""
    x = np.asarray(x, dtype=np.float32)
    kernel = np.asarray(kernel, dtype=np.float32)
    n, c, in_h, in_w = x.shape
    kh, kw = kernel.shape
    p0, p1 = pad
    # zero-stuffing
    z = np.zeros((n, c, in_h * up, in_w * up), dtype=np.float32)
    z[:, :, ::up, ::up] = x
    # pad (negative pads crop)
    z = np.pad(z, ((0, 0), (0, 0), (max(p0, 0), max(p1, 0)), (max(p0, 0), max(p1, 0))))
    z = z[:, :, max(-p0, 0): z.shape[2] - max(-p1, 0), max(-p0, 0): z.shape[3] - max(-p1, 0)]
    oh = z.shape[2] - kh + 1
    ow = z.shape[3] - kw + 1
    kf = kernel[::-1, ::-1]  # true convolution = correlation with the flipped kernel
    out = np.zeros((n, c, oh, ow), dtype=np.float32)
    for a in range(kh):
        for b in range(kw):
            out += kf[a, b] * z[:, :, a:a + oh, b:b + ow]
    return np.ascontiguousarray(out[:, :, ::down, ::down])


def upsample_2d(x, k=(1, 3, 3, 1), factor=2, gain=1):
    """up_or_down_sampling.py:196-225."""
    kk = setup_kernel(k) * (gain * factor ** 2)
    p = kk.shape[0] - factor
    return upfirdn2d(x, kk, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, k=(1, 3, 3, 1), factor=2, gain=1):
    """up_or_down_sampling.py:228-258."""
    kk = setup_kernel(k) * gain
    p = kk.shape[0] - factor
    return upfirdn2d(x, kk, down=factor, pad=((p + 1) // 2, p // 2))
