"""Oracle: entropy-model table logic of compressai==1.1.5 in numpy (test infrastructure only).

PARITY UNPINNED (third-party package absent, no reference fixture).  Restates
`GaussianConditional.{update_scale_table,update,build_indexes}`, `EntropyModel.{quantize,dequantize}`,
`EntropyBottleneck.{_build_indexes,_get_medians}` and the C++ `pmf_to_quantized_cdf`, as used at
reference call sites Network.py:23-27, 307-314, 346-347, 399-401, 423-428, 450, 493-496, 514-517.
"""
import math

import numpy as np
from scipy.stats import norm

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(lo=SCALES_MIN, hi=SCALES_MAX, levels=SCALES_LEVELS):
    """Network.py:23-27 (torch.exp(torch.linspace(log lo, log hi, levels)) in float32)."""
    return np.exp(np.linspace(math.log(lo), math.log(hi), levels, dtype=np.float32)).astype(np.float32)


def pmf_to_quantized_cdf(pmf, precision=16):
    cdf = [0] + [int(np.round(np.float32(p) * np.float32(1 << precision))) for p in pmf]
    total = sum(cdf)
    cdf = [((1 << precision) * p) // total for p in cdf]
    cdf = list(np.cumsum(cdf))
    cdf[-1] = 1 << precision
    n = len(cdf) - 1
    for i in range(n):
        if cdf[i] == cdf[i + 1]:
            best_freq, best = None, -1
            for j in range(n):
                f = cdf[j + 1] - cdf[j]
                if f > 1 and (best_freq is None or f < best_freq):
                    best_freq, best = f, j
            assert best != -1
            if best < i:
                for j in range(best + 1, i + 1):
                    cdf[j] -= 1
            else:
                for j in range(i + 1, best + 1):
                    cdf[j] += 1
    return np.asarray(cdf, dtype=np.int32)


def gaussian_tables(scale_table, tail_mass=1e-9, precision=16):
    """GaussianConditional.update(): (quantized_cdf [n, Lmax+2], cdf_length [n], offset [n])."""
    scale_table = np.asarray(scale_table, dtype=np.float32)
    multiplier = -norm.ppf(tail_mass / 2)
    pmf_center = np.ceil(scale_table * np.float32(multiplier)).astype(np.int32)
    pmf_length = 2 * pmf_center + 1
    max_length = int(pmf_length.max())
    samples = np.abs(np.arange(max_length, dtype=np.int32)[None, :] - pmf_center[:, None]).astype(np.float32)
    s = scale_table[:, None]
    half = np.float32(0.5)
    cdf_fn = lambda v: (half * np.float32(1) * (1 + np.vectorize(math.erf)(v / np.float32(math.sqrt(2))))).astype(np.float32)
    upper = cdf_fn((half - samples) / s)
    lower = cdf_fn((-half - samples) / s)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    qcdf = np.zeros((len(pmf_length), max_length + 2), dtype=np.int32)
    for i in range(len(pmf_length)):
        prob = np.concatenate([pmf[i, :pmf_length[i]], tail[i]])
        c = pmf_to_quantized_cdf(prob, precision)
        qcdf[i, :len(c)] = c
    return qcdf, (pmf_length + 2).astype(np.int32), (-pmf_center).astype(np.int32)


def build_indexes(scales, scale_table):
    """GaussianConditional.build_indexes: LowerBound(0.11) then count table entries below."""
    scales = np.maximum(np.asarray(scales, dtype=np.float32), np.float32(SCALES_MIN))
    idx = np.full(scales.shape, len(scale_table) - 1, dtype=np.int32)
    for s in scale_table[:-1]:
        idx -= (scales <= s).astype(np.int32)
    return idx


def quantize_symbols(x, means):
    """EntropyModel.quantize(x, "symbols", means): round-half-even of x - means, as int32."""
    return np.rint(np.asarray(x, dtype=np.float32) - np.asarray(means, dtype=np.float32)).astype(np.int32)


def logistic_tables(scales, medians, tail_mass=1e-9, precision=16):
    """Synthetic stand-in for EntropyBottleneck.update(): a discretised logistic per channel.
    (The real tables come from the checkpoint; the decoder only ever reads tables + medians.)"""
    scales = np.asarray(scales, dtype=np.float64)
    half_w = np.ceil(scales * math.log(2 / tail_mass - 1)).astype(np.int32) + 1
    pmf_length = 2 * half_w + 1
    max_length = int(pmf_length.max())
    qcdf = np.zeros((len(scales), max_length + 2), dtype=np.int32)
    sig = lambda v: 1 / (1 + np.exp(-v))
    for i, (s, hw) in enumerate(zip(scales, half_w)):
        k = np.arange(-hw, hw + 1, dtype=np.float64)
        pmf = sig((k + 0.5) / s) - sig((k - 0.5) / s)
        tail = max(1 - pmf.sum(), 1e-12)
        c = pmf_to_quantized_cdf(np.concatenate([pmf, [tail]]).astype(np.float32), precision)
        qcdf[i, :len(c)] = c
    return qcdf, (pmf_length + 2).astype(np.int32), (-half_w).astype(np.int32)
