"""Times one 3x3 f16x3 convolution shape through the library at explicit split factors against the default choice."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evc_amd  # noqa
from evc_amd import lib as L
B, R, Ci, Co = (int(v) for v in sys.argv[1:5])
x = torch.randn(B, R, R, Ci, device="cuda")
w = torch.randn(Co, Ci, 3, 3, device="cuda") / (9 * Ci) ** 0.5
wp = L.conv_pack_weights(w, L.ARITH_F16X3)
a, s = torch.ones(B, Ci, device="cuda"), torch.zeros(B, Ci, device="cuda")
fl = 2.0 * B * R * R * Ci * Co * 9
for sp in (1, 0, 2, 0, 1, 3, 0):
    kw = dict(coef=(a, s), act_in=L.ACT_SILU, splits=sp, want_stats=True)
    for _ in range(3): L.conv2d_nhwc(x, wp, Co, 3, 3, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): L.conv2d_nhwc(x, wp, Co, 3, 3, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B} {R}x{R} {Ci}->{Co} splits={sp or 'default'}: {ms*1e3:.1f} us {fl/ms/1e9:.1f} TF/s  ws {L.conv_workspace_bytes(B,R,R,Ci,Co,3,3,splits=sp,arith=L.ARITH_F16X3)>>20} MB", flush=True)
