"""Quick timing probe (not the benchmark): per-layer conv timings and whole-forward time with HIP events."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa: E402,F401
from evc_amd import lib as L  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def conv_probe(B):
    shapes = [(128, 192, 192, 3), (128, 384, 192, 3), (64, 192, 192, 3), (64, 384, 384, 3), (32, 384, 384, 3),
              (32, 576, 576, 3), (16, 576, 576, 3), (8, 768, 768, 3), (8, 1536, 768, 3), (128, 384, 192, 1),
              (32, 384, 1152, 1)]
    for (R, Ci, Co, K) in shapes:
        x = torch.randn(B, R, R, Ci, device="cuda")
        w = L.conv_pack_weights(torch.randn(Co, Ci, K, K, device="cuda") / np.sqrt(Ci * K * K))
        a, s = torch.ones(B, Ci, device="cuda"), torch.zeros(B, Ci, device="cuda")
        flop = 2.0 * B * R * R * Ci * Co * K * K
        t0 = timeit(lambda: L.conv2d_nhwc(x, w, Co, K, K))
        t1 = timeit(lambda: L.conv2d_nhwc(x, w, Co, K, K, coef=(a, s), act_in=L.ACT_SILU))
        print(f"conv B={B} {R}x{R} {Ci}->{Co} k{K}: plain {t0:.3f} ms {flop / t0 / 1e9:.1f} TF/s | "
              f"+gn/silu {t1:.3f} ms {flop / t1 / 1e9:.1f} TF/s", flush=True)


def forward_probe(Bs):
    from evc_amd import synthetic
    from evc_amd.config import default_config as make_config
    from evc_amd.scorenet import ScoreNet
    t0 = time.time()
    cfg = make_config(192, 192, 128)
    net = ScoreNet(cfg, synthetic.diffusion_state_dict(cfg, 1234))
    torch.cuda.synchronize()
    print(f"weights built+packed in {time.time() - t0:.1f}s", flush=True)
    for B in Bs:
        x, c = torch.randn(B, 15, 128, 128, device="cuda"), torch.randn(B, 6, 128, 128, device="cuda")
        for rep in range(2):
            for graphs in (False, True):     # interleaved A/B in one process on one device
                net.use_graphs = graphs
                net.forward_label(x, 500, c)
                torch.cuda.synchronize()
                h0 = time.perf_counter()
                net.forward_label(x, 500, c)
                host_ms = (time.perf_counter() - h0) * 1e3      # host time to enqueue one forward
                t = timeit(lambda: net.forward_label(x, 500, c), iters=3, warm=1)
                print(f"forward B={B} graphs={graphs}: {t:.1f} ms  -> {345.2 * B / t:.1f} TFLOP/s, "
                      f"{t / B:.1f} ms/sample, host enqueue {host_ms:.1f} ms", flush=True)


def layer_probe(B):
    """Per conv shape inside a real forward: launches, time, TFLOP/s (HIP events around every conv launch)."""
    import collections
    from evc_amd import synthetic
    from evc_amd.config import default_config as make_config
    from evc_amd.scorenet import ScoreNet
    cfg = make_config(192, 192, 128)
    net = ScoreNet(cfg, synthetic.diffusion_state_dict(cfg, 1234))
    x, c = torch.randn(B, 15, 128, 128, device="cuda"), torch.randn(B, 6, 128, 128, device="cuda")
    net.forward_label(x, 500, c)
    torch.cuda.synchronize()
    prof = []
    L.CONV_PROFILE = prof
    for _ in range(3):
        net.forward_label(x, 500, c)
    L.CONV_PROFILE = None
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for r in prof:
        k = r["shape"] + (r["split"],)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += r["e0"].elapsed_time(r["e1"]); a[2] += r["flops"]
    tot = sum(a[1] for a in agg.values()) / 3
    print(f"conv total {tot:.2f} ms/forward at B={B}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        Bq, H, W, Ci, Co, K, sp = k
        print(f"  {H:3d}x{W:<3d} {Ci:4d}->{Co:<4d} k{K} splitK={int(sp)} : n={a[0] // 3:2d}  {a[1] / 3:6.3f} ms/fwd "
              f"({100 * a[1] / 3 / tot:4.1f}%)  {a[2] / a[1] / 1e9:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--conv", type=int, default=0)
    ap.add_argument("--forward", type=int, nargs="*", default=[])
    ap.add_argument("--layers", type=int, default=0)
    a = ap.parse_args()
    if a.conv:
        conv_probe(a.conv)
    if a.forward:
        forward_probe(a.forward)
    if a.layers:
        layer_probe(a.layers)
