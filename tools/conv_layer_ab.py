"""Per-launch A/B of a convolution dispatch option inside real forwards (HIP events on the launch stream, side-stream
overlap off): for every distinct conv launch of a B-clip forward, mean time with the option off / on, same process,
alternating rounds.      python tools/conv_layer_ab.py [B] [option] [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa: E402,F401
from evc_amd import lib as L, synthetic  # noqa: E402
from evc_amd.config import default_config  # noqa: E402
from evc_amd.scorenet import ScoreNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
opt = sys.argv[2] if len(sys.argv) > 2 else "wide256"
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cfg = default_config()
net = ScoreNet(cfg, synthetic.diffusion_state_dict(cfg, 1234))
net.overlap_skip = False
x, c = torch.randn(B, 15, 128, 128, device="cuda"), torch.randn(B, 6, 128, 128, device="cuda")
agg = {0: {}, 1: {}}
order = []
reps = 4
for rnd in range(rounds):
    for v in (0, 1):
        L.conv_set_option(opt, v)
        for _ in range(2):
            net.forward_label(x, 500, c)
        torch.cuda.synchronize()
        for _ in range(reps):
            prof = []
            L.CONV_PROFILE = prof
            net.forward_label(x, 500, c)
            L.CONV_PROFILE = None
            torch.cuda.synchronize()
            for r in prof:
                k = r["shape"] + (r["call"]["coef"], r["call"]["res"], r["call"]["x2"], r["arith"])
                a = agg[v].setdefault(k, [0, 0.0, 0.0, r["flops"]])
                if k not in order:
                    order.append(k)
                a[0] += 1
                a[1] += r["e0"].elapsed_time(r["ec"])          # kernel alone
                a[2] += r["e0"].elapsed_time(r["e1"])          # with its split-K combine
tot = {v: sum(a[2] for a in agg[v].values()) / (rounds * reps) for v in (0, 1)}
print(f"B={B}: convolutions per forward {opt}=0: {tot[0]:.3f} ms, {opt}=1: {tot[1]:.3f} ms")
print(" n   B   H   W    Ci    Co K coef res  x2 | off: us kernel  us+combine  TF/s | on: us kernel  us+combine  TF/s | ratio")
for k in sorted(order, key=lambda k: -agg[0][k][2]):
    a0, a1 = agg[0][k], agg[1][k]
    n = a0[0] // (rounds * reps)
    u0, c0, u1, c1 = a0[1] / a0[0] * 1e3, a0[2] / a0[0] * 1e3, a1[1] / a1[0] * 1e3, a1[2] / a1[0] * 1e3
    if abs(c1 / c0 - 1) < 0.01 and n * c0 < 0.01 * tot[0] * 1e3:
        continue
    print(f"{n:2d} {k[0]:3d} {k[1]:3d} {k[2]:3d} {k[3]:5d} {k[4]:5d} {k[5]} {int(k[6])}    {int(k[7])}  {k[8]:4d} | {u0:9.1f} {c0:9.1f} {a0[3] / c0 / 1e6:7.1f} |"
          f" {u1:9.1f} {c1:9.1f} {a1[3] / c1 / 1e6:7.1f} | {c0 / c1:5.3f}")
