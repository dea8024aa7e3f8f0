"""Times the SPADE-conditioned score network at the benchmark batch: map build (first forward of a chunk) vs steady state."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa
from evc_amd.config import default_config
from evc_amd.scorenet import build_score_network
from oracle.scorenet import Dims
from oracle.scorenet_spade import seeded_params
B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
cfg = default_config()
cfg.model.spade = True
d = Dims(ngf=cfg.model.ngf, n_head_channels=cfg.model.n_head_channels, image_size=cfg.data.image_size)
net = build_score_network(cfg, seeded_params(d, 5, spade_dim=128))
x, c = torch.randn(B, 15, 128, 128, device="cuda"), torch.randn(B, 6, 128, 128, device="cuda")
net.forward_label(x, 500, c); torch.cuda.synchronize()
c2 = c.clone()
t0 = time.perf_counter(); net.forward_label(x, 500, c2); torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(3): net.forward_label(x, 499, c2)
torch.cuda.synchronize(); t2 = time.perf_counter()
for _ in range(10): net.forward_label(x, 499, c2)
torch.cuda.synchronize(); t3 = time.perf_counter()
maps = sum(m.numel() * 4 for m in net._maps.values()) / 2**30
print(f"SPADE B={B}: first forward of a chunk (builds {len(net._maps)} gamma/beta maps, {maps:.1f} GiB) {1e3*(t1-t0):.1f} ms; steady forward {1e3*(t3-t2)/10:.2f} ms")
