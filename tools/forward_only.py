"""A few score-network forwards at the benchmark batch size (for rocprofv3 --pmc passes: short, conv-dominated)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa: E402,F401
from evc_amd import synthetic  # noqa: E402
from evc_amd.config import default_config  # noqa: E402
from evc_amd.scorenet import ScoreNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
cfg = default_config()
net = ScoreNet(cfg, synthetic.diffusion_state_dict(cfg, 1234))
x, c = torch.randn(B, 15, 128, 128, device="cuda"), torch.randn(B, 6, 128, 128, device="cuda")
for _ in range(3):
    net.forward_label(x, 500, c)
torch.cuda.synchronize()
print("done")
