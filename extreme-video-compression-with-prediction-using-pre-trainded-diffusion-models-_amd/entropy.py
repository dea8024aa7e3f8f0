"""Host-side entropy-model logic for the ELIC codec, on top of the native range-ANS coder (libevc_rans.so).

Mirrors what the reference reaches through compressai 1.1.5 (Appendix B of SURVEY.md):
``EntropyBottleneck.{compress,decompress}`` for the hyper-latent z and the table builder
``GaussianConditional.update`` (+ C++ ``pmf_to_quantized_cdf``) used when a checkpoint has to be synthesised.
The per-symbol CDF index of the Gaussian-conditional slices is computed on the GPU
(evc_elic_gather_params_f32); only int32 indexes / symbols cross the PCIe link.
"""
import math

import numpy as np
import torch

from . import lib as L

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(lo=SCALES_MIN, hi=SCALES_MAX, levels=SCALES_LEVELS):
    """reference Network.py:23-27."""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


class Tables:
    """Quantised CDF table set of one entropy model: cdf (n, ld) int32, length (n,), offset (n,)."""

    def __init__(self, cdf, length, offset):
        self.cdf = np.ascontiguousarray(np.asarray(cdf, dtype=np.int32))
        self.length = np.ascontiguousarray(np.asarray(length, dtype=np.int32).reshape(-1))
        self.offset = np.ascontiguousarray(np.asarray(offset, dtype=np.int32).reshape(-1))

    @classmethod
    def from_state_dict(cls, sd, prefix):
        return cls(sd[prefix + "._quantized_cdf"].cpu().numpy(), sd[prefix + "._cdf_length"].cpu().numpy(),
                   sd[prefix + "._offset"].cpu().numpy())

    def encode(self, symbols, indexes):
        return L.rans_encode(symbols, indexes, self.cdf, self.length, self.offset)

    def decode(self, data, indexes):
        return L.rans_decode(data, indexes, self.cdf, self.length, self.offset)


def gaussian_conditional_tables(scale_table, tail_mass=1e-9, precision=16):
    """``GaussianConditional.update()``: per scale a discretised zero-mean Gaussian + tail-mass symbol."""
    scale_table = torch.as_tensor(scale_table, dtype=torch.float32)
    # _standardized_quantile(tail_mass / 2) = scipy.stats.norm.ppf
    multiplier = -float(torch.distributions.Normal(0.0, 1.0).icdf(torch.tensor(tail_mass / 2, dtype=torch.float64)))
    pmf_center = torch.ceil(scale_table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(pmf_length.max())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    s = scale_table.unsqueeze(1)
    std_cum = lambda v: 0.5 * torch.erfc(-(2 ** -0.5) * v)
    upper = std_cum((0.5 - samples) / s)
    lower = std_cum((-0.5 - samples) / s)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        prob = torch.cat((pmf[i, :pmf_length[i]], tail[i]), dim=0).numpy()
        c = L.pmf_to_quantized_cdf(prob, precision)
        cdf[i, :len(c)] = c
    return Tables(cdf, (pmf_length + 2).numpy(), (-pmf_center).numpy())


def logistic_bottleneck_tables(scales, tail_mass=1e-9, precision=16):
    """Synthetic stand-in for ``EntropyBottleneck.update()`` (its learned density is not needed to decode:
    the decoder only reads tables + medians, which real checkpoints ship)."""
    scales = np.asarray(scales, dtype=np.float64)
    half_w = np.ceil(scales * math.log(2 / tail_mass - 1)).astype(np.int32) + 1
    pmf_length = 2 * half_w + 1
    cdf = np.zeros((len(scales), int(pmf_length.max()) + 2), dtype=np.int32)
    sig = lambda v: 1 / (1 + np.exp(-v))
    for i, (s, hw) in enumerate(zip(scales, half_w)):
        k = np.arange(-hw, hw + 1, dtype=np.float64)
        pmf = sig((k + 0.5) / s) - sig((k - 0.5) / s)
        tail = max(1 - pmf.sum(), 1e-12)
        c = L.pmf_to_quantized_cdf(np.concatenate([pmf, [tail]]).astype(np.float32), precision)
        cdf[i, :len(c)] = c
    return Tables(cdf, pmf_length + 2, -half_w)


class EntropyBottleneckCodec:
    """z path: index = channel id, symbol = round(z - median) (compressai EntropyBottleneck)."""

    def __init__(self, tables, medians):
        self.tables = tables
        self.medians = np.asarray(medians, dtype=np.float32).reshape(-1)
        self.channels = self.tables.cdf.shape[0]

    def _indexes(self, size):
        return np.broadcast_to(np.arange(self.channels, dtype=np.int32)[:, None, None],
                               (self.channels, size[0], size[1])).reshape(-1)

    def compress(self, z):
        """z: (B, C, h, w) float32 host array -> list of B byte strings."""
        z = np.asarray(z, dtype=np.float32)
        idx = self._indexes(z.shape[-2:])
        out = []
        for b in range(z.shape[0]):
            sym = np.rint(z[b] - self.medians[:, None, None]).astype(np.int32)   # round half to even
            out.append(self.tables.encode(sym.reshape(-1), idx))
        return out

    def decompress(self, strings, size):
        """-> z_hat (B, C, h, w) float32 host array."""
        idx = self._indexes(size)
        out = np.empty((len(strings), self.channels, size[0], size[1]), dtype=np.float32)
        for b, s in enumerate(strings):
            sym = self.tables.decode(s, idx).reshape(self.channels, size[0], size[1])
            out[b] = sym.astype(np.float32) + self.medians[:, None, None]
        return out
