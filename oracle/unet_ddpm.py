"""Oracle: the reference's alternative score network ``UNet_DDPM`` (plain DDPM U-Net), functional torch-CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows reference ``models/unet.py``: ``UNet.__init__`` /
``forward`` (:184-309), ``ResnetBlock`` (:65-99), ``AttnBlock`` (:102-123), ``Nin`` (:48-62), ``Upsample`` (:126-134),
``get_timestep_embedding`` (:148-168), ``UNet_DDPM`` (:335-371).  SURVEY.md section 0: the shipped CLI never builds
this network (it hard-codes the NCSN++ "unetmore"), but it is the file BASELINE.json names and it plugs into the same
samplers (same ``net(x, labels, cond=)`` call, same ``alphas / betas`` buffers), so it is covered as the "alt model".

Parameters: flat dict keyed like the reference ``state_dict()`` (``unet.downblocks.3.conv0.weight`` ...).  Pinned by
``tests/golden/unet_ddpm.npz`` (generated from the imported reference by tests/golden/make_goldens.py).
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Dims:
    ngf: int = 32
    mode: str = "deep"
    channels: int = 3
    num_frames: int = 5
    num_frames_cond: int = 2
    time_conditional: bool = True
    rescaled: bool = True
    logit_transform: bool = False


def ch_mults(d: Dims):
    m = {"deep": (1, 2, 2, 2), "deeper": (1, 2, 2, 4, 4), "deepest": (1, 2, 2, 2, 4, 4)}[d.mode]
    return [d.ngf * n for n in m]


def program(d: Dims):
    """Ordered module records of UNet.__init__ (models/unet.py:214-255): (list name, index, record)."""
    cm = ch_mults(d)
    n_in = d.channels * (d.num_frames + d.num_frames_cond)
    down, mid, up = [], [], []
    down.append(dict(kind="conv", cin=n_in, cout=d.ngf, stride=1))
    prev, ch_size = cm[0], [d.ngf]
    for i, ich in enumerate(cm):
        for first in (prev, ich):
            down.append(dict(kind="res", cin=first, cout=ich))
            ch_size.append(ich)
            if i == 1:
                down.append(dict(kind="attn", ch=ich))
        if i != len(cm) - 1:
            down.append(dict(kind="conv", cin=ich, cout=ich, stride=2))
            ch_size.append(ich)
        prev = ich
    mid += [dict(kind="res", cin=cm[-1], cout=cm[-1]), dict(kind="attn", ch=cm[-1]), dict(kind="res", cin=cm[-1], cout=cm[-1])]
    prev = cm[-1]
    for i, ich in reversed(list(enumerate(cm))):
        for _ in range(3):
            skip = ch_size.pop()
            up.append(dict(kind="res", cin=prev + skip, cout=ich, split=(prev, skip)))
            if i == 1:
                up.append(dict(kind="attn", ch=ich))
            prev = ich
        if i != 0:
            up.append(dict(kind="upsample", ch=ich))
    assert not ch_size
    return [("downblocks", j, m) for j, m in enumerate(down)] + [("middleblocks", j, m) for j, m in enumerate(mid)] + \
           [("upblocks", j, m) for j, m in enumerate(up)]


def param_shapes(d: Dims, prefix="unet."):
    out = []
    tdim = 4 * d.ngf
    for lst, j, m in program(d):
        n = f"{prefix}{lst}.{j}"
        if m["kind"] == "conv":
            out += [(n + ".weight", (m["cout"], m["cin"], 3, 3)), (n + ".bias", (m["cout"],))]
        elif m["kind"] == "upsample":
            out += [(n + ".conv.weight", (m["ch"], m["ch"], 3, 3)), (n + ".conv.bias", (m["ch"],))]
        elif m["kind"] == "res":
            ci, co = m["cin"], m["cout"]
            out += [(n + ".normalize0.weight", (ci,)), (n + ".normalize0.bias", (ci,)),
                    (n + ".conv0.weight", (co, ci, 3, 3)), (n + ".conv0.bias", (co,))]
            if d.time_conditional:
                out += [(n + ".dense.weight", (co, tdim)), (n + ".dense.bias", (co,))]
            out += [(n + ".normalize1.weight", (co,)), (n + ".normalize1.bias", (co,)),
                    (n + ".conv1.weight", (co, co, 3, 3)), (n + ".conv1.bias", (co,))]
            if ci != co:
                out += [(n + ".nin.weights", (co, ci)), (n + ".nin.bias", (co,))]
        elif m["kind"] == "attn":
            c = m["ch"]
            for nm in ("Q", "K", "V", "OUT"):
                out += [(f"{n}.{nm}.weights", (c, c)), (f"{n}.{nm}.bias", (c,))]
            out += [(n + ".normalize.weight", (c,)), (n + ".normalize.bias", (c,))]
    out += [(prefix + "normalize.weight", (d.ngf,)), (prefix + "normalize.bias", (d.ngf,)),
            (prefix + "out.weight", (d.channels * d.num_frames, d.ngf, 3, 3)), (prefix + "out.bias", (d.channels * d.num_frames,)),
            (prefix + "temb_dense.0.weight", (tdim, d.ngf)), (prefix + "temb_dense.0.bias", (tdim,)),
            (prefix + "temb_dense.2.weight", (tdim, tdim)), (prefix + "temb_dense.2.bias", (tdim,))]
    return out


def seeded_params(d: Dims, seed, prefix="unet."):
    """numpy default_rng(seed) normals / sqrt(fan_in) (the reference's init zeroes conv1 / OUT / out: degenerate)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes(d, prefix):
        leaf = name.rsplit(".", 1)[1]
        if leaf == "bias":
            a = 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif "normalize" in name and leaf == "weight":
            a = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        else:
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(int(np.prod(shape[1:]))))
        p[name] = torch.from_numpy(a)
    return p


def timestep_embedding(t, dim):
    """models/unet.py:148-168."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = t.float()[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], 1)


def _swish(x):
    return x * torch.sigmoid(x)


def _gn(x, p, n):
    return F.group_norm(x, 32, p[n + ".weight"], p[n + ".bias"], eps=1e-6)


def _nin(x, p, n):
    return torch.einsum("oc,bchw->bohw", p[n + ".weights"], x) + p[n + ".bias"][None, :, None, None]


def _res(x, temb, p, n, d):
    h = F.conv2d(_swish(_gn(x, p, n + ".normalize0")), p[n + ".conv0.weight"], p[n + ".conv0.bias"], padding=1)
    if temb is not None and d.time_conditional:
        h = h + F.linear(temb, p[n + ".dense.weight"], p[n + ".dense.bias"])[:, :, None, None]
    h = _swish(_gn(h, p, n + ".normalize1"))
    skip = _nin(x, p, n + ".nin") if (n + ".nin.weights") in p else x
    return skip + F.conv2d(h, p[n + ".conv1.weight"], p[n + ".conv1.bias"], padding=1)


def _attn(x, p, n):
    B, C, H, W = x.shape
    h = _gn(x, p, n + ".normalize")
    q, k, v = (_nin(h, p, f"{n}.{m}").reshape(B, C, H * W) for m in ("Q", "K", "V"))
    w = torch.softmax(torch.einsum("bcq,bck->bqk", q, k) / math.sqrt(C), dim=-1)
    o = torch.einsum("bqk,bck->bcq", w, v).reshape(B, C, H, W)
    return x + _nin(o, p, n + ".OUT")


@torch.no_grad()
def forward(p, d: Dims, x, labels, cond=None, prefix="unet."):
    """UNet.forward (models/unet.py:258-309) behind UNet_DDPM.forward (noise_in_cond off)."""
    temb = None
    if labels is not None and d.time_conditional:
        temb = timestep_embedding(labels, d.ngf)
        temb = _swish(F.linear(temb, p[prefix + "temb_dense.0.weight"], p[prefix + "temb_dense.0.bias"]))
        temb = _swish(F.linear(temb, p[prefix + "temb_dense.2.weight"], p[prefix + "temb_dense.2.bias"]))
    if cond is not None:
        x = torch.cat([x, cond], 1)
    if not d.logit_transform and not d.rescaled:
        x = 2 * x - 1.0
    hs = []
    for lst, j, m in program(d):
        n = f"{prefix}{lst}.{j}"
        if lst == "upblocks" and m["kind"] == "res":
            x = torch.cat((x, hs.pop()), 1)
        if m["kind"] == "conv":
            x = F.conv2d(x, p[n + ".weight"], p[n + ".bias"], stride=m["stride"], padding=1)
        elif m["kind"] == "res":
            x = _res(x, temb, p, n, d)
        elif m["kind"] == "attn":
            x = _attn(x, p, n)
        elif m["kind"] == "upsample":
            x = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), p[n + ".conv.weight"], p[n + ".conv.bias"], padding=1)
        if lst == "downblocks":
            if m["kind"] == "attn":
                hs.pop()
            hs.append(x)
    assert not hs
    x = _swish(_gn(x, p, prefix + "normalize"))
    return F.conv2d(x, p[prefix + "out.weight"], p[prefix + "out.bias"], padding=1)
