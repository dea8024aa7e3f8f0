"""Per-launch table of the convolutions of one B-clip forward (HIP events on the launch stream, side-stream overlap
off): shape, mean time over a few forwards, TFLOP/s -- to set beside tools/conv_bench.hip's isolated numbers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa: E402,F401
from evc_amd import lib as L, synthetic  # noqa: E402
from evc_amd.config import default_config  # noqa: E402
from evc_amd.scorenet import ScoreNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
cfg = default_config()
net = ScoreNet(cfg, synthetic.diffusion_state_dict(cfg, 1234))
net.overlap_skip = False
x, c = torch.randn(B, 15, 128, 128, device="cuda"), torch.randn(B, 6, 128, 128, device="cuda")
for _ in range(2):
    net.forward_label(x, 500, c)
torch.cuda.synchronize()
reps = 5
runs = []
for _ in range(reps):
    prof = []
    L.CONV_PROFILE = prof
    net.forward_label(x, 500, c)
    L.CONV_PROFILE = None
    torch.cuda.synchronize()
    runs.append(prof)
agg = {}
order = []
for j, r in enumerate(runs[0]):
    ms = sum(run[j]["e0"].elapsed_time(run[j]["e1"]) for run in runs) / reps
    k = r["shape"] + (r["call"]["coef"], r["call"]["res"], r["arith"], r["split"])
    if k not in agg:
        agg[k] = [0, 0.0, r["flops"]]
        order.append(k)
    agg[k][0] += 1
    agg[k][1] += ms
tot = sum(v[1] for v in agg.values())
print(f"B={B}: {len(runs[0])} conv launches, {tot:.3f} ms per forward in convolutions")
print(" n   B   H   W    Ci    Co K coef res arith slabs |  ms each   TF/s   % of conv time")
for k in sorted(order, key=lambda k: -agg[k][1]):
    n, ms, fl = agg[k]
    print(f"{n:2d} {k[0]:3d} {k[1]:3d} {k[2]:3d} {k[3]:5d} {k[4]:5d} {k[5]} {int(k[6])}    {int(k[7])}   {k[8]}     {int(k[9])}    | {ms / n:8.4f} {fl / (ms / n) / 1e9:7.1f}  {100 * ms / tot:5.1f}")
