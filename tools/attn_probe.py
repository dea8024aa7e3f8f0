"""Times evc_attention_f32 on the three attention shapes of the score network (B clips)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa
from evc_amd import lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for (R, C, heads) in ((32, 384, 2), (16, 576, 3), (8, 768, 4)):
    N = R * R
    qkv = torch.randn(B, N, 3 * C, device="cuda")
    bounds = torch.zeros(3, dtype=torch.int32, device="cuda")
    L.moments_bound(L.chan_stats(qkv.view(B, 1, N, 3 * C)), 0, C, bounds)
    fl = 4.0 * N * N * (C // heads) * B * heads
    for name, bd in (("f32 mfma", None), ("f16x3", bounds)):
        for _ in range(3): L.attention(qkv, C, heads, bounds=bd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): L.attention(qkv, C, heads, bounds=bd)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"B={B} {R}x{R} C={C} heads={heads} {name:8s}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TF/s", flush=True)
