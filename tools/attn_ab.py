"""Same-process A/B of the attention option "kv_planes" at the score network's attention shapes (B clips):
    python tools/attn_ab.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa: E402,F401
from evc_amd import lib as L  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 9
for (N, heads, D) in ((1024, 2, 192), (256, 3, 192), (64, 4, 192)):
    C = heads * D
    qkv = torch.randn(B, N, 3 * C, device="cuda")
    st = L.chan_stats(qkv.view(B, N, 1, 3 * C))
    bounds = torch.zeros(3, dtype=torch.int32, device="cuda")
    L.moments_bound(st, 0, C, bounds)
    res = {}
    for rnd in range(3):
        for v in (0, 1):
            L.attention_set_option("kv_planes", v)
            for _ in range(3):
                L.attention(qkv, C, heads, bounds=bounds)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                L.attention(qkv, C, heads, bounds=bounds)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
    flops = 4.0 * B * heads * N * N * D
    for v in (0, 1):
        us = min(res[v])
        print(f"B={B} N={N} heads={heads} D={D} kv_planes={v}: {us:7.1f} us per call (incl. pre-pass / merge)  {flops / us / 1e6:6.1f} TFLOP/s   all: {[round(x, 1) for x in res[v]]}")
