"""Achieved HBM rate of the memory-bound kernel classes at the benchmark's shapes (B = 9), as algorithmic bytes
(each tensor read / written once) over the HIP-event time: FIR resampling with GroupNorm+SiLU on load, per-channel
moments, the stand-alone affine+activation pass, the split-K combine, the sampler step, layout pack / unpack.
SURVEY.md 8d asks for these beside the MFMA roofline of the convolution."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa: E402,F401
from evc_amd import lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
PEAK = 8000.0


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def report(name, nbytes, us):
    gbs = nbytes / us / 1e3
    print(f"{name:58s} {nbytes / 1e6:8.1f} MB  {us:8.1f} us  {gbs:7.0f} GB/s  ({gbs / PEAK:.2f} of 8 TB/s)", flush=True)


k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32) / 64
for (R, C) in ((128, 192), (64, 384)):
    x = torch.randn(B, R, R, C, device="cuda")
    a, s = torch.rand(B, C, device="cuda") + 0.5, torch.randn(B, C, device="cuda")
    n = x.numel() * 4
    report(f"upfirdn2d_nhwc down {R}->{R // 2} C={C} (+GN/SiLU on load)", n + n // 4,
           timeit(lambda: L.upfirdn2d_nhwc(x, k, 1, 2, (1, 1), coef=(a, s), act=L.ACT_SILU)))
    xs = torch.randn(B, R // 2, R // 2, C, device="cuda")
    report(f"upfirdn2d_nhwc up {R // 2}->{R} C={C} (+GN/SiLU on load)", xs.numel() * 4 + n,
           timeit(lambda: L.upfirdn2d_nhwc(xs, k * 4, 2, 1, (2, 1), coef=(a, s), act=L.ACT_SILU)))
    report(f"upfirdn2d_nhwc down {R}->{R // 2} C={C} (plain)", n + n // 4, timeit(lambda: L.upfirdn2d_nhwc(x, k, 1, 2, (1, 1))))
    report(f"upfirdn2d_nhwc up {R // 2}->{R} C={C} (plain)", xs.numel() * 4 + n, timeit(lambda: L.upfirdn2d_nhwc(xs, k * 4, 2, 1, (2, 1))))
    report(f"chan_stats {R}x{R} C={C}", n, timeit(lambda: L.chan_stats(x)))
    report(f"affine_act {R}x{R} C={C}", 2 * n, timeit(lambda: L.affine_act(x, (a, s), L.ACT_SILU)))
# split-K combine: the conv call minus the same conv without the combine is not separable from Python; time the
# whole split launch and report the combine's algorithmic bytes against rocprofv3's per-kernel average instead
for (R, Ci, Co, S) in ((64, 192, 192, 3), (32, 384, 384, 7)):
    M = B * R * R
    print(f"conv_splitk_reduce {R}x{R} {Ci}->{Co} splits={S}: algorithmic {(S + 2) * M * Co * 4 / 1e6:.1f} MB "
          f"({S} slabs + residual read, 1 write); see profiles/*kernel_stats.csv for its duration")
x = torch.randn(B, 15, 128, 128, device="cuda")
e, nz = torch.randn_like(x), torch.randn_like(x)
report("ddpm_step (x, eps, noise -> x)", 4 * x.numel() * 4, timeit(lambda: L.ddpm_step(x, e, nz, 1.0, 0.1, 0.5, 0.5, 0.1, True)))
c = torch.randn(B, 6, 128, 128, device="cuda")
report("pack_nchw_to_nhwc (15+6 -> 32 ch)", (21 + 32) * B * 128 * 128 * 4, timeit(lambda: L.pack_nchw_to_nhwc(x, c, 32)))
