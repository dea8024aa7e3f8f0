import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import rnd
import evc_amd
from evc_amd import sampler as S, lib as L
from evc_amd.config import default_config
from evc_amd.decoder import ClipDecoder
from evc_amd.scorenet import ScoreNet
from oracle import scorenet as ON
cfg = default_config(32, 32, 128, subsample=2)
net = ScoreNet(cfg, ON.seeded_params(ON.Dims(ngf=32, n_head_channels=32, image_size=128), 3))
dec = ClipDecoder(net, None, cfg, S.get_sampler("DDPM"))
cond = rnd(60, 7, 2, 3, 128, 128).clamp(-1, 1).add(1).div(2).cuda()
def noise_fn(tag, shp):
    return rnd(hash(str(tag)) % 1000 + 7, 7, *shp[1:])[:shp[0]]
full = dec.generate(cond, noise_fn=noise_fn, groups=1)
for b in range(7):
    nf = lambda tag, shp, b=b: rnd(hash(str(tag)) % 1000 + 7, 7, *shp[1:])[b:b+1]
    one = dec.generate(cond[b:b+1].contiguous(), noise_fn=nf, groups=1)
    print(b, float((one[0] - full[b]).abs().max()), float((one[0]-full[b]).abs().mean()))
# forward-level check
x = rnd(1, 7, 15, 128, 128).cuda(); c = rnd(2, 7, 6, 128, 128).cuda()
o7 = net.forward_label(x, 500, c)
for b in (0, 3, 6):
    o1 = net.forward_label(x[b:b+1].contiguous(), 500, c[b:b+1].contiguous())
    print("fwd", b, float((o1[0]-o7[b]).abs().max() / o7[b].abs().max()))
