"""Oracle: ELIC key-frame codec (TestModel) as a functional torch-CPU restatement -- test infrastructure only.

PARITY UNPINNED: reference ``Network.py`` cannot be imported here (needs timm, compressai, thop, ptflops --
ordinary ModuleNotFoundError) and the reference holds no fixture for this path, so this file is written
from the source text: ``Network.py:33-59`` (ResidualBottleneckBlock), ``:74-170`` (layer stacks),
``:336-441`` (compress), ``:444-532`` (decompress), ``ELICUtilis/layers/layers.py:64-88``
(CheckboardMaskedConv2d), ``:202-253`` (AttentionBlock), ``Inference.py:19-75`` (pad / crop / bit count),
with compressai 1.1.5 ``conv`` / ``deconv`` helpers (k=5, stride 2, padding 2, output_padding 1) and the
entropy-model behaviour restated in oracle/entropy.py + oracle/rans.py.

Parameters: flat dict with the reference's state-dict names (``g_s.1.weight``,
``g_s.0.conv_a.0.conv.0.weight``, ``context_prediction.2.mask``, ``gaussian_conditional._quantized_cdf`` ...).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import entropy as E
from . import exact as X
from . import rans as R

GROUPS = [0, 16, 16, 32, 64, 192]   # Network.py:87
N, M = 192, 320


def _conv(p, n, x, stride=1, pad=0):
    return F.conv2d(x, p[n + ".weight"], p[n + ".bias"], stride=stride, padding=pad)


def _deconv(p, n, x):
    """compressai deconv(): ConvTranspose2d(k=5, stride=2, output_padding=1, padding=2)."""
    return F.conv_transpose2d(x, p[n + ".weight"], p[n + ".bias"], stride=2, padding=2, output_padding=1)


def _rbb(p, n, x):
    """ResidualBottleneckBlock.forward, Network.py:48-59."""
    out = F.relu(_conv(p, n + ".conv1", x))
    out = F.relu(_conv(p, n + ".conv2", out, pad=1))
    return _conv(p, n + ".conv3", out) + x


def _res_unit(p, n, x):
    """AttentionBlock.ResidualUnit.forward, layers.py:230-235."""
    out = F.relu(_conv(p, n + ".conv.0", x))
    out = F.relu(_conv(p, n + ".conv.2", out, pad=1))
    out = _conv(p, n + ".conv.4", out)
    return F.relu(out + x)


def _attention(p, n, x):
    """AttentionBlock.forward, layers.py:246-253: a * sigmoid(b) + x."""
    a = x
    for i in range(3):
        a = _res_unit(p, f"{n}.conv_a.{i}", a)
    b = x
    for i in range(3):
        b = _res_unit(p, f"{n}.conv_b.{i}", b)
    b = _conv(p, n + ".conv_b.3", b)
    return a * torch.sigmoid(b) + x


def g_a(p, x):
    """Network.py:88-104."""
    x = _conv(p, "g_a.0", x, 2, 2)
    for i in (1, 2, 3):
        x = _rbb(p, f"g_a.{i}", x)
    x = _conv(p, "g_a.4", x, 2, 2)
    for i in (5, 6, 7):
        x = _rbb(p, f"g_a.{i}", x)
    x = _attention(p, "g_a.8", x)
    x = _conv(p, "g_a.9", x, 2, 2)
    for i in (10, 11, 12):
        x = _rbb(p, f"g_a.{i}", x)
    x = _conv(p, "g_a.13", x, 2, 2)
    return _attention(p, "g_a.14", x)


def g_s(p, y):
    """Network.py:106-122."""
    x = _attention(p, "g_s.0", y)
    x = _deconv(p, "g_s.1", x)
    for i in (2, 3, 4):
        x = _rbb(p, f"g_s.{i}", x)
    x = _deconv(p, "g_s.5", x)
    x = _attention(p, "g_s.6", x)
    for i in (7, 8, 9):
        x = _rbb(p, f"g_s.{i}", x)
    x = _deconv(p, "g_s.10", x)
    for i in (11, 12, 13):
        x = _rbb(p, f"g_s.{i}", x)
    return _deconv(p, "g_s.14", x)


def h_a(p, y):
    """Network.py:124-130."""
    x = F.relu(_conv(p, "h_a.0", y, pad=1))
    x = F.relu(_conv(p, "h_a.2", x, 2, 2))
    return _conv(p, "h_a.4", x, 2, 2)


def h_s(p, z, exact=False):
    """Network.py:132-138.  ``exact``: the product's fp32 arithmetic bit for bit (oracle/exact.py) instead of torch's."""
    if exact:
        x = X.deconv5x5s2(z, p["h_s.0.weight"], p["h_s.0.bias"], relu_out=True)
        x = X.deconv5x5s2(x, p["h_s.2.weight"], p["h_s.2.bias"], relu_out=True)
        return X.conv2d(x, p["h_s.4.weight"], p["h_s.4.bias"])
    x = F.relu(_deconv(p, "h_s.0", z))
    x = F.relu(_deconv(p, "h_s.2", x))
    return _conv(p, "h_s.4", x, pad=1)


def _cc(p, i, x, exact=False):
    """cc_transforms[i], Network.py:140-149 (5x5 stride-1 convs, padding 2)."""
    n = f"cc_transforms.{i}"
    if exact:
        x = X.conv2d(x, p[n + ".0.weight"], p[n + ".0.bias"], relu_out=True)
        x = X.conv2d(x, p[n + ".2.weight"], p[n + ".2.bias"], relu_out=True)
        return X.conv2d(x, p[n + ".4.weight"], p[n + ".4.bias"])
    x = F.relu(_conv(p, n + ".0", x, pad=2))
    x = F.relu(_conv(p, n + ".2", x, pad=2))
    return _conv(p, n + ".4", x, pad=2)


def _ctx(p, i, x, exact=False):
    """CheckboardMaskedConv2d.forward, layers.py:84-88: weight * mask, 5x5, padding 2."""
    n = f"context_prediction.{i}"
    if exact:
        return X.conv2d(x, p[n + ".weight"] * p[n + ".mask"], p[n + ".bias"])
    return F.conv2d(x, p[n + ".weight"] * p[n + ".mask"], p[n + ".bias"], padding=2)


def _pa(p, i, x, exact=False):
    """ParamAggregation[i], Network.py:157-166."""
    n = f"ParamAggregation.{i}"
    if exact:
        x = X.conv2d(x, p[n + ".0.weight"], p[n + ".0.bias"], relu_out=True)
        x = X.conv2d(x, p[n + ".2.weight"], p[n + ".2.bias"], relu_out=True)
        return X.conv2d(x, p[n + ".4.weight"], p[n + ".4.bias"])
    x = F.relu(_conv(p, n + ".0", x))
    x = F.relu(_conv(p, n + ".2", x))
    return _conv(p, n + ".4", x)


def checkerboard_mask(shape):
    m = torch.zeros(shape)
    m[:, :, 0::2, 1::2] = 1
    m[:, :, 1::2, 0::2] = 1
    return m


def _pack(t, parity):
    """(B,C,H,W) -> (B,C,H,W/2): anchors (parity 0) = even rows even cols / odd rows odd cols (Network.py:388-393)."""
    B, C, H, W = t.shape
    out = torch.zeros(B, C, H, W // 2)
    o0, o1 = (0, 1) if parity == 0 else (1, 0)
    out[:, :, 0::2, :] = t[:, :, 0::2, o0::2]
    out[:, :, 1::2, :] = t[:, :, 1::2, o1::2]
    return out


def _unpack(t, parity, W):
    B, C, H, _ = t.shape
    out = torch.zeros(B, C, H, W)
    o0, o1 = (0, 1) if parity == 0 else (1, 0)
    out[:, :, 0::2, o0::2] = t[:, :, 0::2, :]
    out[:, :, 1::2, o1::2] = t[:, :, 1::2, :]
    return out


def _tables(p, prefix):
    return (p[prefix + "._quantized_cdf"].numpy().astype(np.int32), p[prefix + "._cdf_length"].numpy().astype(np.int32),
            p[prefix + "._offset"].numpy().astype(np.int32))


def _support(p, i, y_hat_slices, latent_means, latent_scales, exact=False):
    if i == 0:
        return torch.cat([latent_means, latent_scales], dim=1)
    sup = y_hat_slices[0] if i == 1 else torch.cat([y_hat_slices[0], y_hat_slices[i - 1]], dim=1)
    cc = _cc(p, i - 1, sup, exact)
    cc_mean, cc_scale = cc.chunk(2, 1)
    return torch.cat([cc_mean, cc_scale, latent_means, latent_scales], dim=1)


@torch.no_grad()
def compress(p, x, coder=R, exact=False):
    """TestModel.compress, Network.py:336-441 -> {"strings": [y_strings, z_strings], "shape"}; also returns y_hat.
    ``exact``: the entropy-parameter networks (h_s, cc_transforms, context_prediction, ParamAggregation) in the product's
    fp32 arithmetic bit for bit, so that the stream decodes on the HIP codec and vice versa (oracle/exact.py)."""
    y = g_a(p, x)
    B, C, H, W = y.shape
    z = h_a(p, y)
    cdf, cdf_len, off = _tables(p, "entropy_bottleneck")
    med = p["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    zc = z.shape[1]
    z_idx = np.broadcast_to(np.arange(zc, dtype=np.int32)[:, None, None], z.shape[1:]).reshape(-1)
    z_strings, z_hat = [], torch.zeros_like(z)
    for b in range(B):
        sym = E.quantize_symbols(z[b].numpy(), med[0].expand_as(z[b]).numpy()).reshape(-1)
        s = coder.encode_with_indexes(sym.tolist(), z_idx.tolist(), cdf, cdf_len, off)
        z_strings.append(s)
        dec = np.asarray(coder.decode_with_indexes(s, z_idx.tolist(), cdf, cdf_len, off), dtype=np.float32)
        z_hat[b] = torch.from_numpy(dec.reshape(z.shape[1:])) + med[0]
    latent_means, latent_scales = h_s(p, z_hat, exact).chunk(2, 1)
    gcdf, gcdf_len, goff = _tables(p, "gaussian_conditional")
    table = p["gaussian_conditional.scale_table"].numpy()
    y_slices = torch.split(y, GROUPS[1:], 1)
    y_strings, y_hat_slices = [], []
    for i, y_slice in enumerate(y_slices):
        g = GROUPS[i + 1]
        support = _support(p, i, y_hat_slices, latent_means, latent_scales, exact)
        strings_i = []
        y_hat_i = torch.zeros_like(y_slice)
        ctx = torch.zeros(B, 2 * g, H, W)
        for parity in (0, 1):
            if parity == 1:
                ctx = _ctx(p, i, y_hat_i, exact)        # anchors decoded, non-anchor sites still zero
            means, scales = _pa(p, i, torch.cat([ctx, support], dim=1), exact).chunk(2, 1)
            m_enc, s_enc, y_enc = _pack(means, parity), _pack(scales, parity), _pack(y_slice, parity)
            idx = E.build_indexes(s_enc.numpy(), table)
            q = torch.zeros_like(m_enc)
            strs = []
            for b in range(B):
                sym = E.quantize_symbols(y_enc[b].numpy(), m_enc[b].numpy()).reshape(-1)
                s = coder.encode_with_indexes(sym.tolist(), idx[b].reshape(-1).tolist(), gcdf, gcdf_len, goff)
                strs.append(s)
                dec = np.asarray(coder.decode_with_indexes(s, idx[b].reshape(-1).tolist(), gcdf, gcdf_len, goff))
                assert np.array_equal(dec, sym)
                q[b] = torch.from_numpy(dec.astype(np.float32).reshape(m_enc[b].shape)) + m_enc[b]
            strings_i.append(strs)
            y_hat_i = y_hat_i + _unpack(q, parity, W)
        y_strings.append(strings_i)
        y_hat_slices.append(y_hat_i)
    return {"strings": [y_strings, z_strings], "shape": z.shape[-2:], "y_hat": torch.cat(y_hat_slices, 1), "y": y}


@torch.no_grad()
def decompress(p, strings, shape, coder=R, return_latents=False, exact=False):
    """TestModel.decompress, Network.py:444-532.  ``exact``: see ``compress``."""
    y_strings, z_strings = strings
    B = len(z_strings)
    cdf, cdf_len, off = _tables(p, "entropy_bottleneck")
    zc = cdf.shape[0]
    med = p["entropy_bottleneck.quantiles"][:, 0, 1].reshape(-1, 1, 1)
    z_idx = np.broadcast_to(np.arange(zc, dtype=np.int32)[:, None, None], (zc, shape[0], shape[1])).reshape(-1)
    z_hat = torch.zeros(B, zc, shape[0], shape[1])
    for b in range(B):
        dec = np.asarray(coder.decode_with_indexes(z_strings[b], z_idx.tolist(), cdf, cdf_len, off), dtype=np.float32)
        z_hat[b] = torch.from_numpy(dec.reshape(zc, shape[0], shape[1])) + med
    latent_means, latent_scales = h_s(p, z_hat, exact).chunk(2, 1)
    H, W = z_hat.shape[2] * 4, z_hat.shape[3] * 4
    gcdf, gcdf_len, goff = _tables(p, "gaussian_conditional")
    table = p["gaussian_conditional.scale_table"].numpy()
    y_hat_slices, all_symbols = [], []
    for i in range(len(GROUPS) - 1):
        g = GROUPS[i + 1]
        support = _support(p, i, y_hat_slices, latent_means, latent_scales, exact)
        y_hat_i = torch.zeros(B, g, H, W)
        ctx = torch.zeros(B, 2 * g, H, W)
        for parity in (0, 1):
            if parity == 1:
                ctx = _ctx(p, i, y_hat_i, exact)
            means, scales = _pa(p, i, torch.cat([ctx, support], dim=1), exact).chunk(2, 1)
            m_enc, s_enc = _pack(means, parity), _pack(scales, parity)
            idx = E.build_indexes(s_enc.numpy(), table)
            q = torch.zeros_like(m_enc)
            for b in range(B):
                dec = np.asarray(coder.decode_with_indexes(y_strings[i][parity][b], idx[b].reshape(-1).tolist(),
                                                           gcdf, gcdf_len, goff))
                all_symbols.append(dec.astype(np.int32))
                q[b] = torch.from_numpy(dec.astype(np.float32).reshape(m_enc[b].shape)) + m_enc[b]
            y_hat_i = y_hat_i + _unpack(q, parity, W)
        y_hat_slices.append(y_hat_i)
    y_hat = torch.cat(y_hat_slices, dim=1)
    x_hat = g_s(p, y_hat).clamp_(0, 1)
    out = {"x_hat": x_hat}
    if return_latents:
        out.update(y_hat=y_hat, z_hat=z_hat, symbols=all_symbols)
    return out


def count_bits(strings):
    """Inference.py:51-67: 8 * total byte length of every string in the nested list."""
    total = 0
    for s in strings:
        for j in s:
            if isinstance(j, list):
                for i in j:
                    total += sum(len(k) for k in i) if isinstance(i, list) else len(i)
            else:
                total += len(j)
    return 8 * total


@torch.no_grad()
def inference(p, x, patch=64, coder=R):
    """Inference.inference, Inference.py:19-75: x (3,H,W) in [0,1] -> (x_hat (1,3,H,W), bits)."""
    x = x.unsqueeze(0)
    h, w = x.size(2), x.size(3)
    new_h, new_w = (h + patch - 1) // patch * patch, (w + patch - 1) // patch * patch
    xp = F.pad(x, (0, new_w - w, 0, new_h - h))
    enc = compress(p, xp, coder)
    dec = decompress(p, enc["strings"], enc["shape"], coder)
    x_hat = F.pad(dec["x_hat"], (0, -(new_w - w), 0, -(new_h - h)))
    return x_hat, count_bits(enc["strings"])
