"""Forward-level same-process A/B of a convolution dispatch option (default: "wide256", the 256 x 192 one-workgroup-per-CU
kernel): alternating rounds of whole score-network forwards at the benchmark batch size, HIP-event timed.

    python tools/ab_forward.py [B] [option] [rounds] [forwards per round]
"""
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
nfwd = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cfg = default_config()
net = ScoreNet(cfg, synthetic.diffusion_state_dict(cfg, 1234))
x, c = torch.randn(B, 15, 128, 128, device="cuda"), torch.randn(B, 6, 128, 128, device="cuda")
outs = {}
for r in range(rounds):
    for v in (0, 1):
        L.conv_set_option(opt, v)
        for _ in range(2):
            o = net.forward_label(x, 500, c)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nfwd):
            o = net.forward_label(x, 500, c)
        e1.record()
        torch.cuda.synchronize()
        outs[v] = o
        print(f"round {r} {opt}={v}: {e0.elapsed_time(e1) / nfwd:.3f} ms per B={B} forward", flush=True)
d = float((outs[0] - outs[1]).abs().max() / outs[0].abs().max())
print(f"max |out({opt}=0) - out({opt}=1)| / max|out| = {d:.2e}; range events {L.range_events()}")
