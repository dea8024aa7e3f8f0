"""Oracle: NCSN++ "unetmore" score network forward, functional torch-CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows reference
``models/better/ncsnpp_more.py:32-392,721-770`` (module list + forward),
``models/better/layerspp.py:207-249`` (AttnBlockpp), ``:486-549`` (get_act_norm),
``:553-624`` (ResnetBlockBigGANppGN), ``models/better/layers.py:504-518,535-544``
(timestep embedding, NIN) and ``models/better/up_or_down_sampling.py:196-258`` (FIR resampling).

Parameters are a flat ``dict`` keyed exactly like the reference ``state_dict()``
(``unet.all_modules.<i>.<name>``), so a reference checkpoint or the seeded weights of
``tests/golden/make_goldens.py`` can be used unchanged.
"""
import math
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Dims:
    ngf: int = 192
    ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 3, 4])
    num_res_blocks: int = 2
    attn_resolutions: List[int] = field(default_factory=lambda: [8, 16, 32])
    n_head_channels: int = 192
    image_size: int = 128
    channels: int = 3
    num_frames: int = 5
    num_frames_cond: int = 2
    cond_emb: bool = False          # config.model.cond_emb: Embedding(2, ngf // 2) appended to the time embedding


def num_groups(ch):
    """layerspp.py:473-476 / :212-214."""
    g = min(ch // 4, 32)
    while ch % g != 0:
        g -= 1
    return g


def program(d: Dims):
    """The ordered module list of NCSNpp.__init__ (ncsnpp_more.py:70-247), as plain records."""
    mods = [dict(kind="linear", cin=d.ngf, cout=4 * d.ngf), dict(kind="linear", cin=4 * d.ngf, cout=4 * d.ngf)]
    res = [d.image_size // (2 ** i) for i in range(len(d.ch_mult))]
    n_in = d.channels * (d.num_frames + d.num_frames_cond)
    mods.append(dict(kind="conv3", cin=n_in, cout=d.ngf))
    hs_c = [d.ngf]
    in_ch = d.ngf
    for lvl, mult in enumerate(d.ch_mult):
        for _ in range(d.num_res_blocks):
            out_ch = d.ngf * mult
            mods.append(dict(kind="res", cin=in_ch, cout=out_ch, up=False, down=False))
            in_ch = out_ch
            if res[lvl] in d.attn_resolutions:
                mods.append(dict(kind="attn", ch=in_ch))
            hs_c.append(in_ch)
        if lvl != len(d.ch_mult) - 1:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=True))
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False))
    mods.append(dict(kind="attn", ch=in_ch))
    mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=False, down=False))
    for lvl in reversed(range(len(d.ch_mult))):
        for _ in range(d.num_res_blocks + 1):
            out_ch = d.ngf * d.ch_mult[lvl]
            mods.append(dict(kind="res", cin=in_ch + hs_c.pop(), cout=out_ch, up=False, down=False))
            in_ch = out_ch
        if res[lvl] in d.attn_resolutions:
            mods.append(dict(kind="attn", ch=in_ch))
        if lvl != 0:
            mods.append(dict(kind="res", cin=in_ch, cout=in_ch, up=True, down=False))
    assert not hs_c
    mods.append(dict(kind="norm", ch=in_ch))
    mods.append(dict(kind="conv3", cin=in_ch, cout=d.channels * d.num_frames))
    return mods


def timestep_embedding(timesteps, dim, max_positions=10000):
    """layers.py:504-518."""
    half = dim // 2
    emb = math.log(max_positions) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = timesteps.float()[:, None] * emb[None, :]
    return torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)


def _fir_kernel():
    k = np.outer([1, 3, 3, 1], [1, 3, 3, 1]).astype(np.float32)
    return torch.from_numpy(k / k.sum())


def fir_up2(x):
    """upsample_2d(x, [1,3,3,1], factor=2): zero-stuff x2, pad (2,1), 4x4 FIR with gain 4."""
    b, c, h, w = x.shape
    z = x.new_zeros(b, c, 2 * h, 2 * w)
    z[:, :, ::2, ::2] = x
    z = F.pad(z, (2, 1, 2, 1))
    k = (_fir_kernel() * 4).flip(0, 1)[None, None].repeat(c, 1, 1, 1)
    return F.conv2d(z, k, groups=c)


def fir_down2(x):
    """downsample_2d(x, [1,3,3,1], factor=2): pad (1,1), 4x4 FIR, keep every 2nd sample."""
    c = x.shape[1]
    z = F.pad(x, (1, 1, 1, 1))
    k = _fir_kernel().flip(0, 1)[None, None].repeat(c, 1, 1, 1)
    return F.conv2d(z, k, groups=c, stride=2)


def _adagn_silu(x, temb, p, prefix):
    """get_act_norm.forward (layerspp.py:518-549), emb branch."""
    c = x.shape[1]
    emb = F.linear(F.silu(temb), p[prefix + ".Dense_0.weight"], p[prefix + ".Dense_0.bias"])[:, :, None, None]
    scale, shift = torch.chunk(emb, 2, dim=1)
    y = F.group_norm(x, num_groups(c), None, None, 1e-5)
    return F.silu(y * (1 + scale) + shift)


def _resblock(x, temb, p, pre, m):
    """ResnetBlockBigGANppGN.forward (layerspp.py:595-624)."""
    h = _adagn_silu(x, temb, p, pre + ".actnorm0")
    if m["up"]:
        h, x = fir_up2(h), fir_up2(x)
    elif m["down"]:
        h, x = fir_down2(h), fir_down2(x)
    h = F.conv2d(h, p[pre + ".Conv_0.weight"], p[pre + ".Conv_0.bias"], padding=1)
    h = _adagn_silu(h, temb, p, pre + ".actnorm1")
    h = F.conv2d(h, p[pre + ".Conv_1.weight"], p[pre + ".Conv_1.bias"], padding=1)
    if m["cin"] != m["cout"] or m["up"] or m["down"]:
        x = F.conv2d(x, p[pre + ".Conv_2.weight"], p[pre + ".Conv_2.bias"])
    return (x + h) / np.sqrt(2.)


def _nin(x, w, b):
    """layers.py:541-544: per-pixel x @ W + b with W stored (in, out)."""
    return torch.einsum("bchw,cd->bdhw", x, w) + b[None, :, None, None]


def _attn(x, p, pre, head_ch):
    """AttnBlockpp.forward (layerspp.py:230-249)."""
    B, C, H, W = x.shape
    heads = 1 if C < head_ch else C // head_ch
    h = F.group_norm(x, num_groups(C), p[pre + ".GroupNorm_0.weight"], p[pre + ".GroupNorm_0.bias"], 1e-6)
    q = _nin(h, p[pre + ".NIN_0.W"], p[pre + ".NIN_0.b"]).reshape(B * heads, C // heads, H * W)
    k = _nin(h, p[pre + ".NIN_1.W"], p[pre + ".NIN_1.b"]).reshape(B * heads, C // heads, H * W)
    v = _nin(h, p[pre + ".NIN_2.W"], p[pre + ".NIN_2.b"]).reshape(B * heads, C // heads, H * W)
    w = torch.einsum("bcq,bck->bqk", q, k) * (int(C // heads) ** (-0.5))
    w = F.softmax(w, dim=-1)
    o = torch.einsum("bqk,bck->bcq", w, v).reshape(B, C, H, W)
    o = _nin(o, p[pre + ".NIN_3.W"], p[pre + ".NIN_3.b"])
    return (x + o) / np.sqrt(2.)


@torch.no_grad()
def noise_cond(cond, labels, alphas, z):
    """UNetMore_DDPM.forward with noise_in_cond (ncsnpp_more.py:755-768): the conditioning frames noised to each sample's
    level, ``z`` the draw (Gaussian; a standardised Gamma draw on gamma models)."""
    a = alphas[labels.long()].reshape(cond.shape[0], *([1] * len(cond.shape[1:])))
    return a.sqrt() * cond + (1 - a).sqrt() * z


def forward(p, d: Dims, x, labels, cond=None, prefix="unet.all_modules.", taps=None, cond_mask=None):
    """UNetMore_DDPM.forward -> NCSNpp.forward (ncsnpp_more.py:753-770, :251-392).

    ``taps``: optional dict; if given, the output of every module index is stored in it (program indices)."""
    mods = program(d)
    name = lambda i: prefix + str(_sd_index(d, i))
    if cond is not None:
        x = torch.cat([x, cond], dim=1)
    temb = timestep_embedding(labels, d.ngf)
    temb = F.linear(temb, p[name(0) + ".weight"], p[name(0) + ".bias"])
    temb = F.linear(F.silu(temb), p[name(1) + ".weight"], p[name(1) + ".bias"])
    if d.cond_emb:                                                                   # :282-285
        if cond_mask is None:
            cond_mask = torch.ones(x.shape[0], dtype=torch.int32)
        temb = torch.cat([temb, F.embedding(cond_mask.long(), p[prefix + "2.weight"])], dim=1)
    i = 2
    x = x.contiguous().float()

    def run(i, h):
        m = mods[i]
        if m["kind"] == "res":
            out = _resblock(h, temb, p, name(i), m)
        elif m["kind"] == "attn":
            out = _attn(h, p, name(i), d.n_head_channels)
        else:
            raise AssertionError(m)
        if taps is not None:
            taps[i] = out
        return out

    hs = [F.conv2d(x, p[name(i) + ".weight"], p[name(i) + ".bias"], padding=1)]
    if taps is not None:
        taps[i] = hs[0]
    i += 1
    n_lvl = len(d.ch_mult)
    for lvl in range(n_lvl):
        for _ in range(d.num_res_blocks):
            h = run(i, hs[-1]); i += 1
            if h.shape[-1] in d.attn_resolutions:
                h = run(i, h); i += 1
            hs.append(h)
        if lvl != n_lvl - 1:
            h = run(i, hs[-1]); i += 1
            hs.append(h)
    h = hs[-1]
    h = run(i, h); i += 1
    h = run(i, h); i += 1
    h = run(i, h); i += 1
    for lvl in reversed(range(n_lvl)):
        for _ in range(d.num_res_blocks + 1):
            h = run(i, torch.cat([h, hs.pop()], dim=1)); i += 1
        if h.shape[-1] in d.attn_resolutions:
            h = run(i, h); i += 1
        if lvl != 0:
            h = run(i, h); i += 1
    assert not hs
    c = h.shape[1]
    h = F.silu(F.group_norm(h, num_groups(c), p[name(i) + ".Norm_0.weight"], p[name(i) + ".Norm_0.bias"], 1e-5))
    i += 1
    h = F.conv2d(h, p[name(i) + ".weight"], p[name(i) + ".bias"], padding=1)
    i += 1
    assert i == len(mods)
    return h


def _sd_index(d, i):
    """State-dict index of program entry ``i``: with cond_emb the Embedding sits at modules[2] and shifts the rest."""
    return i + (1 if d.cond_emb and i >= 2 else 0)


def param_shapes(d: Dims, prefix="unet.all_modules."):
    """Ordered (name, shape) list in reference ``state_dict()`` order for the network parameters."""
    out = []
    td = 4 * d.ngf + (d.ngf // 2 if d.cond_emb else 0)           # temb_dim (ncsnpp_more.py:95-99)
    for i, m in enumerate(program(d)):
        n = prefix + str(_sd_index(d, i))
        if i == 2 and d.cond_emb:
            out += [(prefix + "2.weight", (2, d.ngf // 2))]      # torch.nn.Embedding(2, nf // 2), modules[2]
        if m["kind"] == "linear":
            out += [(n + ".weight", (m["cout"], m["cin"])), (n + ".bias", (m["cout"],))]
        elif m["kind"] == "conv3":
            out += [(n + ".weight", (m["cout"], m["cin"], 3, 3)), (n + ".bias", (m["cout"],))]
        elif m["kind"] == "res":
            ci, co = m["cin"], m["cout"]
            out += [(n + ".actnorm0.Dense_0.weight", (2 * ci, td)), (n + ".actnorm0.Dense_0.bias", (2 * ci,)),
                    (n + ".Conv_0.weight", (co, ci, 3, 3)), (n + ".Conv_0.bias", (co,)),
                    (n + ".actnorm1.Dense_0.weight", (2 * co, td)), (n + ".actnorm1.Dense_0.bias", (2 * co,)),
                    (n + ".Conv_1.weight", (co, co, 3, 3)), (n + ".Conv_1.bias", (co,))]
            if ci != co or m["up"] or m["down"]:
                out += [(n + ".Conv_2.weight", (co, ci, 1, 1)), (n + ".Conv_2.bias", (co,))]
        elif m["kind"] == "attn":
            c = m["ch"]
            out += [(n + ".GroupNorm_0.weight", (c,)), (n + ".GroupNorm_0.bias", (c,))]
            for j in range(4):
                out += [(n + f".NIN_{j}.W", (c, c)), (n + f".NIN_{j}.b", (c,))]
        elif m["kind"] == "norm":
            out += [(n + ".Norm_0.weight", (m["ch"],)), (n + ".Norm_0.bias", (m["ch"],))]
    return out


def seeded_params(d: Dims, seed, prefix="unet.all_modules."):
    """Framework-independent weights: numpy default_rng(seed) normals scaled by 1/sqrt(fan_in).

    The reference's own init is degenerate (1e-10 variance on every block's last layer,
    layers.py:77-80), so goldens re-randomise every parameter with this recipe; the same function
    rebuilds them in tests, which keeps the 1 GB tensor out of the repository."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes(d, prefix):
        leaf = name.rsplit(".", 1)[1]
        if leaf in ("bias", "b"):
            a = 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif "Norm_0.weight" in name or "GroupNorm_0.weight" in name:
            a = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif leaf == "W":
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(shape[0]))
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(fan_in))
        p[name] = torch.from_numpy(a)
    return p
