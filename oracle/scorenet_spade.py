"""Oracle: the SPADE variant of the NCSN++ score network (``model.spade: true``), functional torch-CPU restatement.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows reference ``models/better/ncsnpp_more.py:396-718``
(``SPADE_NCSNpp``: module list + forward), ``models/better/layerspp.py:101-173`` (``MySPADE``), ``:486-549``
(``get_act_norm``, ``norm == 'spade'``) and ``:628-705`` (``ResnetBlockBigGANppSPADE``).  Differences from the
concat-conditioned network of oracle/scorenet.py: the conditioning frames are NOT concatenated to the input (the first
convolution sees ``channels * num_frames`` channels); every act-norm is

    SiLU( [ GroupNorm_noaffine(x, eps 1e-6) * (1 + gamma(cond)) + beta(cond) ] * (1 + scale(t)) + shift(t) )

with ``gamma / beta = conv3x3(SiLU(conv3x3(nearest_resize(cond))))`` per act-norm (``spade_dim`` hidden channels); the
final norm is the same without the time embedding.  SURVEY.md section 2 row 4b: out of scope for the shipped config
(``mine.yml:117`` has ``spade: false``), covered as an alt model (section 8 row f4).

Parameters: flat dict keyed like the reference ``state_dict()``.  Pinned by ``tests/golden/forward_spade.npz``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .scorenet import Dims, _attn, fir_down2, fir_up2, num_groups, timestep_embedding


def program(d: Dims):
    """Module list of SPADE_NCSNpp.__init__ (ncsnpp_more.py:436-586): same order as NCSNpp, other first conv."""
    from .scorenet import program as base
    mods = base(d)
    assert mods[2]["kind"] == "conv3"
    mods[2] = dict(kind="conv3", cin=d.channels * d.num_frames, cout=d.ngf)
    return mods


def _spade(x, cond, p, pre):
    """MySPADE.forward (layerspp.py:152-173), 2-D case."""
    c = x.shape[1]
    normalized = F.group_norm(x, num_groups(c), None, None, 1e-6)
    seg = F.interpolate(cond, size=x.shape[-2:], mode="nearest")
    actv = F.silu(F.conv2d(seg, p[pre + ".mlp_shared.0.weight"], p[pre + ".mlp_shared.0.bias"], padding=1))
    gamma = F.conv2d(actv, p[pre + ".mlp_gamma.weight"], p[pre + ".mlp_gamma.bias"], padding=1)
    beta = F.conv2d(actv, p[pre + ".mlp_beta.weight"], p[pre + ".mlp_beta.bias"], padding=1)
    return normalized * (1 + gamma) + beta


def _actnorm(x, temb, cond, p, pre):
    """get_act_norm.forward with norm == 'spade' (layerspp.py:518-549)."""
    y = _spade(x, cond, p, pre + ".Norm_0")
    if temb is not None:
        emb = F.linear(F.silu(temb), p[pre + ".Dense_0.weight"], p[pre + ".Dense_0.bias"])[:, :, None, None]
        scale, shift = torch.chunk(emb, 2, dim=1)
        y = y * (1 + scale) + shift
    return F.silu(y)


def _resblock(x, temb, cond, p, pre, m):
    """ResnetBlockBigGANppSPADE.forward (layerspp.py:675-705)."""
    h = _actnorm(x, temb, cond, p, pre + ".actnorm0")
    if m["up"]:
        h, x = fir_up2(h), fir_up2(x)
    elif m["down"]:
        h, x = fir_down2(h), fir_down2(x)
    h = F.conv2d(h, p[pre + ".Conv_0.weight"], p[pre + ".Conv_0.bias"], padding=1)
    h = _actnorm(h, temb, cond, p, pre + ".actnorm1")
    h = F.conv2d(h, p[pre + ".Conv_1.weight"], p[pre + ".Conv_1.bias"], padding=1)
    if m["cin"] != m["cout"] or m["up"] or m["down"]:
        x = F.conv2d(x, p[pre + ".Conv_2.weight"], p[pre + ".Conv_2.bias"])
    return (x + h) / np.sqrt(2.)


@torch.no_grad()
def forward(p, d: Dims, x, labels, cond, prefix="unet.all_modules.", spade_dim=128):
    """UNetMore_DDPM.forward -> SPADE_NCSNpp.forward (ncsnpp_more.py:590-718)."""
    mods = program(d)
    name = lambda i: prefix + str(i)
    temb = timestep_embedding(labels, d.ngf)
    temb = F.linear(temb, p[name(0) + ".weight"], p[name(0) + ".bias"])
    temb = F.linear(F.silu(temb), p[name(1) + ".weight"], p[name(1) + ".bias"])
    x = x.contiguous().float()
    cond = cond.float()
    i = 2

    def run(i, h):
        m = mods[i]
        if m["kind"] == "res":
            return _resblock(h, temb, cond, p, name(i), m)
        if m["kind"] == "attn":
            return _attn(h, p, name(i), d.n_head_channels)
        raise AssertionError(m)

    hs = [F.conv2d(x, p[name(i) + ".weight"], p[name(i) + ".bias"], padding=1)]
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
    h = _actnorm(h, None, cond, p, name(i))
    i += 1
    h = F.conv2d(h, p[name(i) + ".weight"], p[name(i) + ".bias"], padding=1)
    i += 1
    assert i == len(mods)
    return h


def _spade_shapes(n, ch, cond_ch, spade_dim):
    return [(n + ".mlp_shared.0.weight", (spade_dim, cond_ch, 3, 3)), (n + ".mlp_shared.0.bias", (spade_dim,)),
            (n + ".mlp_gamma.weight", (ch, spade_dim, 3, 3)), (n + ".mlp_gamma.bias", (ch,)),
            (n + ".mlp_beta.weight", (ch, spade_dim, 3, 3)), (n + ".mlp_beta.bias", (ch,))]


def param_shapes(d: Dims, prefix="unet.all_modules.", spade_dim=128):
    """(name, shape) in the reference ``state_dict()`` order (per act-norm: Dense_0, then Norm_0.mlp_*)."""
    cond_ch = d.channels * d.num_frames_cond
    out = []
    for i, m in enumerate(program(d)):
        n = prefix + str(i)
        if m["kind"] == "linear":
            out += [(n + ".weight", (m["cout"], m["cin"])), (n + ".bias", (m["cout"],))]
        elif m["kind"] == "conv3":
            out += [(n + ".weight", (m["cout"], m["cin"], 3, 3)), (n + ".bias", (m["cout"],))]
        elif m["kind"] == "res":
            ci, co = m["cin"], m["cout"]
            out += [(n + ".actnorm0.Dense_0.weight", (2 * ci, 4 * d.ngf)), (n + ".actnorm0.Dense_0.bias", (2 * ci,))]
            out += _spade_shapes(n + ".actnorm0.Norm_0", ci, cond_ch, spade_dim)
            out += [(n + ".Conv_0.weight", (co, ci, 3, 3)), (n + ".Conv_0.bias", (co,)),
                    (n + ".actnorm1.Dense_0.weight", (2 * co, 4 * d.ngf)), (n + ".actnorm1.Dense_0.bias", (2 * co,))]
            out += _spade_shapes(n + ".actnorm1.Norm_0", co, cond_ch, spade_dim)
            out += [(n + ".Conv_1.weight", (co, co, 3, 3)), (n + ".Conv_1.bias", (co,))]
            if ci != co or m["up"] or m["down"]:
                out += [(n + ".Conv_2.weight", (co, ci, 1, 1)), (n + ".Conv_2.bias", (co,))]
        elif m["kind"] == "attn":
            c = m["ch"]
            out += [(n + ".GroupNorm_0.weight", (c,)), (n + ".GroupNorm_0.bias", (c,))]
            for j in range(4):
                out += [(n + f".NIN_{j}.W", (c, c)), (n + f".NIN_{j}.b", (c,))]
        elif m["kind"] == "norm":
            out += _spade_shapes(n + ".Norm_0", m["ch"], cond_ch, spade_dim)
    return out


def seeded_params(d: Dims, seed, prefix="unet.all_modules.", spade_dim=128):
    """Same recipe as oracle/scorenet.py:seeded_params (default_rng normals / sqrt(fan_in); biases 0.1 sigma)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes(d, prefix, spade_dim):
        leaf = name.rsplit(".", 1)[1]
        if leaf in ("bias", "b"):
            a = 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif "GroupNorm_0.weight" in name:
            a = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif leaf == "W":
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(shape[0]))
        else:
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(int(np.prod(shape[1:]))))
        p[name] = torch.from_numpy(a)
    return p
