"""Synthetic stand-ins for the assets the reference downloads (none of them is available offline):
reference-layout checkpoints with seeded random weights, and smooth random 30-frame clips.

* diffusion: ``checkpoints/sender/checkpoint_<id>.pt`` layout (list, ``module.`` prefix, EMA shadow) -- see ckpt.py;
  every parameter ~ N(0, 1/fan_in) from ``numpy.random.default_rng(seed)`` in state-dict order (the
  reference's own init is degenerate: 1e-10 variance on each block's last layer, models/better/layers.py:77-80);
* ELIC: flat state dict with the key names of ``TestModel`` (Network.py:74-170) incl. compressai's
  entropy-model buffers, whose tables are built the way ``GaussianConditional.update`` builds them;
* clips: ``uint8 (n, 30, 3, 128, 128)`` low-pass filtered noise with slow motion, shaped like ``city_bonn.npy``.
"""
import math

import numpy as np
import torch

from . import entropy
from .scorenet import build_program, dims_from_config


# ---- diffusion ------------------------------------------------------------------------------------
def pseudo3d_param_shapes(config, prefix="unet.all_modules."):
    """State-dict layout of NCSNpp with arch = unetmorepseudo3d / unetmore3d (ncsnpp_more.py:70-247 is3d branches, layers3d.py)."""
    from .scorenet_pseudo3d import build_program_3d
    d = dims_from_config(config)
    t = 4 * d.ngf * (d.num_frames + d.num_frames_cond)
    out = []

    def pconv(n, co, ci, k):
        if config.model.arch == "unetmore3d":            # MyConv3d (layers3d.py:225-254)
            return [(n + ".conv.weight", (co, ci, k, k, k)), (n + ".conv.bias", (co,))]
        return [(n + ".space_conv.weight", (co, ci, k, k)), (n + ".space_conv.bias", (co,)),
                (n + ".time_conv.weight", (co, co, k)), (n + ".time_conv.bias", (co,))]
    for i, m in enumerate(build_program_3d(d)):
        n = prefix + str(i)
        k = m["kind"]
        if k == "linear":
            out += [(n + ".weight", (t, t // 4 if i == 0 else t)), (n + ".bias", (t,))]
        elif k in ("conv_in", "conv_out"):
            out += pconv(n, m["cout"], m["cin"], 3)
        elif k == "mix":
            out += [(n + ".weight", (m["frames_out"], m["frames"], 1, 1)), (n + ".bias", (m["frames_out"],))]
        elif k == "res":
            ci, co = m["cin"], m["cout"]
            out += [(n + ".actnorm0.Dense_0.weight", (2 * ci, t)), (n + ".actnorm0.Dense_0.bias", (2 * ci,))]
            out += pconv(n + ".Conv_0", co, ci, 3)
            out += [(n + ".actnorm1.Dense_0.weight", (2 * co, t)), (n + ".actnorm1.Dense_0.bias", (2 * co,))]
            out += pconv(n + ".Conv_1", co, co, 3)
            if ci != co or m["up"] or m["down"]:
                out += pconv(n + ".Conv_2", co, ci, 1)
        elif k == "attn":
            c = m["ch"]
            for part in ("space_att", "time_att"):
                out += [(f"{n}.{part}.GroupNorm_0.weight", (c,)), (f"{n}.{part}.GroupNorm_0.bias", (c,))]
                for j in range(4):
                    out += [(f"{n}.{part}.NIN_{j}.W", (c, c)), (f"{n}.{part}.NIN_{j}.b", (c,))]
        elif k == "norm":
            out += [(n + ".Norm_0.weight", (m["ch"],)), (n + ".Norm_0.bias", (m["ch"],))]
    return out


def diffusion_param_shapes(config, prefix="unet.all_modules."):
    if getattr(config.model, "arch", "unetmore") in ("unetmorepseudo3d", "unetmore3d"):
        return pseudo3d_param_shapes(config, prefix)
    d = dims_from_config(config)
    out = []
    t = 4 * d.ngf
    spade = bool(getattr(config.model, "spade", False))      # SPADE_NCSNpp (ncsnpp_more.py:396-586): per-act-norm gamma / beta convs
    sdim = getattr(config.model, "spade_dim", 128)
    cond_ch = d.channels * d.num_frames_cond

    def spade_shapes(n, ch):
        return [(n + ".mlp_shared.0.weight", (sdim, cond_ch, 3, 3)), (n + ".mlp_shared.0.bias", (sdim,)),
                (n + ".mlp_gamma.weight", (ch, sdim, 3, 3)), (n + ".mlp_gamma.bias", (ch,)),
                (n + ".mlp_beta.weight", (ch, sdim, 3, 3)), (n + ".mlp_beta.bias", (ch,))]
    program = build_program(d)
    if spade:
        program[2]["cin"] = d.channels * d.num_frames
    for i, m in enumerate(program):
        n = prefix + str(i)
        k = m["kind"]
        if k == "linear":
            cin = d.ngf if i == 0 else t
            out += [(n + ".weight", (t, cin)), (n + ".bias", (t,))]
        elif k in ("conv_in", "conv_out"):
            out += [(n + ".weight", (m["cout"], m["cin"], 3, 3)), (n + ".bias", (m["cout"],))]
        elif k == "res":
            ci, co = m["cin"], m["cout"]
            out += [(n + ".actnorm0.Dense_0.weight", (2 * ci, t)), (n + ".actnorm0.Dense_0.bias", (2 * ci,))]
            out += spade_shapes(n + ".actnorm0.Norm_0", ci) if spade else []
            out += [(n + ".Conv_0.weight", (co, ci, 3, 3)), (n + ".Conv_0.bias", (co,)),
                    (n + ".actnorm1.Dense_0.weight", (2 * co, t)), (n + ".actnorm1.Dense_0.bias", (2 * co,))]
            out += spade_shapes(n + ".actnorm1.Norm_0", co) if spade else []
            out += [(n + ".Conv_1.weight", (co, co, 3, 3)), (n + ".Conv_1.bias", (co,))]
            if ci != co or m["up"] or m["down"]:
                out += [(n + ".Conv_2.weight", (co, ci, 1, 1)), (n + ".Conv_2.bias", (co,))]
        elif k == "attn":
            c = m["ch"]
            out += [(n + ".GroupNorm_0.weight", (c,)), (n + ".GroupNorm_0.bias", (c,))]
            for j in range(4):
                out += [(n + f".NIN_{j}.W", (c, c)), (n + f".NIN_{j}.b", (c,))]
        elif k == "norm":
            out += spade_shapes(n + ".Norm_0", m["ch"]) if spade else \
                [(n + ".Norm_0.weight", (m["ch"],)), (n + ".Norm_0.bias", (m["ch"],))]
    return out


def diffusion_state_dict(config, seed):
    rng = np.random.default_rng(seed)
    sd = {}
    for name, shape in diffusion_param_shapes(config):
        leaf = name.rsplit(".", 1)[1]
        if leaf in ("bias", "b"):
            a = 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif name.endswith("Norm_0.weight"):   # GroupNorm_0.weight / Norm_0.weight (not Norm_0.mlp_*.weight)
            a = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
        elif leaf == "W":
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(shape[0]))
        else:
            a = rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(int(np.prod(shape[1:]))))
        sd[name] = torch.from_numpy(a)
    return sd


# ---- ELIC -------------------------------------------------------------------------------------------
def elic_param_shapes(N=192, M=320):
    G = [0, 16, 16, 32, 64, 192]
    out = []

    def conv(n, ci, co, k):
        out.extend([(n + ".weight", (co, ci, k, k)), (n + ".bias", (co,))])

    def deconv(n, ci, co, k=5):
        out.extend([(n + ".weight", (ci, co, k, k)), (n + ".bias", (co,))])

    def rbb(n, c):
        conv(n + ".conv1", c, c // 2, 1); conv(n + ".conv2", c // 2, c // 2, 3); conv(n + ".conv3", c // 2, c, 1)

    def attn(n, c):
        for br in ("conv_a", "conv_b"):
            for i in range(3):
                conv(f"{n}.{br}.{i}.conv.0", c, c // 2, 1); conv(f"{n}.{br}.{i}.conv.2", c // 2, c // 2, 3)
                conv(f"{n}.{br}.{i}.conv.4", c // 2, c, 1)
        conv(n + ".conv_b.3", c, c, 1)
    # g_a (Network.py:88-104)
    conv("g_a.0", 3, N, 5)
    for i in (1, 2, 3):
        rbb(f"g_a.{i}", N)
    conv("g_a.4", N, N, 5)
    for i in (5, 6, 7):
        rbb(f"g_a.{i}", N)
    attn("g_a.8", N)
    conv("g_a.9", N, N, 5)
    for i in (10, 11, 12):
        rbb(f"g_a.{i}", N)
    conv("g_a.13", N, M, 5)
    attn("g_a.14", M)
    # g_s (Network.py:106-122)
    attn("g_s.0", M)
    deconv("g_s.1", M, N)
    for i in (2, 3, 4):
        rbb(f"g_s.{i}", N)
    deconv("g_s.5", N, N)
    attn("g_s.6", N)
    for i in (7, 8, 9):
        rbb(f"g_s.{i}", N)
    deconv("g_s.10", N, N)
    for i in (11, 12, 13):
        rbb(f"g_s.{i}", N)
    deconv("g_s.14", N, 3)
    # hyper transforms (Network.py:124-138)
    conv("h_a.0", M, N, 3); conv("h_a.2", N, N, 5); conv("h_a.4", N, N, 5)
    deconv("h_s.0", N, N); deconv("h_s.2", N, N * 3 // 2); conv("h_s.4", N * 3 // 2, 2 * M, 3)
    for i in range(1, 5):   # Network.py:140-149
        cin = G[1] + (G[i] if i > 1 else 0)
        conv(f"cc_transforms.{i - 1}.0", cin, 224, 5); conv(f"cc_transforms.{i - 1}.2", 224, 128, 5)
        conv(f"cc_transforms.{i - 1}.4", 128, 2 * G[i + 1], 5)
    for i in range(5):      # Network.py:151-166
        conv(f"context_prediction.{i}", G[i + 1], 2 * G[i + 1], 5)
        cin = 640 + G[i + 1 if i > 0 else 0] * 2 + G[i + 1] * 2
        conv(f"ParamAggregation.{i}.0", cin, 640, 1); conv(f"ParamAggregation.{i}.2", 640, 512, 1)
        conv(f"ParamAggregation.{i}.4", 512, 2 * G[i + 1], 1)
    return out


def elic_state_dict(seed, N=192, M=320, gain=1.0):
    """Random ELIC checkpoint in the reference layout.  ParamAggregation scale outputs get a positive bias so
    the coded symbols are mostly in-table (the bypass path still occurs, but is not the only one exercised)."""
    rng = np.random.default_rng(seed)
    G = [0, 16, 16, 32, 64, 192]
    sd = {}
    for name, shape in elic_param_shapes(N, M):
        if name.endswith(".weight"):
            deconv = name[:-7] in ("g_s.1", "g_s.5", "g_s.10", "g_s.14", "h_s.0", "h_s.2")
            fan_in = shape[0 if deconv else 1] * shape[2] * shape[3]
            if deconv:
                fan_in /= 4.0   # each output sees ~1/4 of the taps of a stride-2 transposed conv
            a = gain * rng.standard_normal(shape, dtype=np.float32) / np.float32(math.sqrt(fan_in))
        else:
            a = 0.05 * rng.standard_normal(shape, dtype=np.float32)
        sd[name] = torch.from_numpy(a)
    for i in range(5):
        g = G[i + 1]
        sd[f"ParamAggregation.{i}.4.bias"][g:] += 1.2      # scale half
        mask = torch.zeros(2 * g, g, 5, 5)
        mask[:, :, 0::2, 1::2] = 1
        mask[:, :, 1::2, 0::2] = 1
        sd[f"context_prediction.{i}.mask"] = mask
    table = entropy.get_scale_table()
    gc = entropy.gaussian_conditional_tables(table)
    sd["gaussian_conditional.scale_table"] = table
    sd["gaussian_conditional._quantized_cdf"] = torch.from_numpy(gc.cdf.copy())
    sd["gaussian_conditional._cdf_length"] = torch.from_numpy(gc.length.copy())
    sd["gaussian_conditional._offset"] = torch.from_numpy(gc.offset.copy())
    eb_scales = 0.5 + 2.5 * rng.random(N)
    eb = entropy.logistic_bottleneck_tables(eb_scales)
    med = 0.3 * rng.standard_normal(N).astype(np.float32)
    q = np.stack([med - 10, med, med + 10], 1)[:, None, :].astype(np.float32)
    sd["entropy_bottleneck.quantiles"] = torch.from_numpy(q)
    sd["entropy_bottleneck._quantized_cdf"] = torch.from_numpy(eb.cdf.copy())
    sd["entropy_bottleneck._cdf_length"] = torch.from_numpy(eb.length.copy())
    sd["entropy_bottleneck._offset"] = torch.from_numpy(eb.offset.copy())
    return sd


# ---- clips ------------------------------------------------------------------------------------------
def make_clips(n, seed=0, frames=30, size=128):
    """uint8 (n, frames, 3, size, size): low-pass filtered noise drifting a little per frame."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, frames, 3, size, size), dtype=np.uint8)
    k = np.exp(-0.5 * (np.arange(-8, 9) / 3.0) ** 2)
    k /= k.sum()
    for i in range(n):
        base = rng.standard_normal((3, size + frames, size + frames)).astype(np.float32)
        for ax in (1, 2):
            base = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), ax, base)
        base = (base - base.min()) / (base.max() - base.min() + 1e-9)
        for t in range(frames):
            out[i, t] = np.clip(base[:, t:t + size, t // 2:t // 2 + size] * 255.0, 0, 255).astype(np.uint8)
    return out
