"""Oracle: LPIPS v0.1 with the AlexNet backbone (test infrastructure only; parity UNPINNED).

The sender's decision rule is built on ``lpips.LPIPS(net='alex')`` (reference ``city_sender.py:302``, called per frame by
``decide_5to5_lpips``, ``:376-406``, on the [0, 1] frames as they are -- no ``normalize=True``).  The metric is third-party:
``lpips==0.1.4`` (``requirements.txt:66``) on ``torchvision.models.alexnet``; neither package is importable here and the
reference tree holds no LPIPS fixture, so this file restates the PUBLISHED algorithm of those packages and nothing pins it:

* ``ScalingLayer``: ``(x - shift) / scale`` with shift (-.030, -.088, -.188), scale (.458, .448, .450);
* AlexNet ``features``: conv 11x11 s4 p2 (3->64), ReLU | MaxPool 3 s2, conv 5x5 p2 (64->192), ReLU | MaxPool 3 s2,
  conv 3x3 p1 (192->384), ReLU | conv 3x3 p1 (384->256), ReLU | conv 3x3 p1 (256->256), ReLU -- the five ReLU outputs are taps;
* per tap: ``normalize_tensor`` (x / (sqrt(sum_c x^2) + 1e-10)), squared difference, ``NetLinLayer`` (1x1 conv, one output
  channel, no bias; dropout is inactive in eval), spatial mean; the distance is the sum over the taps.

State-dict key names follow the two packages: ``features.{0,3,6,8,10}.{weight,bias}`` (torchvision AlexNet) and
``lin{0..4}.model.1.weight`` of shape (1, C, 1, 1) (lpips ``weights/v0.1/alex.pth``).
"""
import numpy as np
import torch
import torch.nn.functional as F

SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)
CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1), (10, 256, 256, 3, 1, 1))
CHANNELS = (64, 192, 384, 256, 256)


def seeded_state_dict(seed):
    """Stand-in weights in the packages' layouts (the real ones cannot be fetched offline): He-scaled normal convolutions,
    small biases, non-negative lin weights as the trained ones are."""
    rng = np.random.default_rng(seed)
    sd = {}
    for idx, ci, co, k, _, _ in CONVS:
        sd[f"features.{idx}.weight"] = torch.from_numpy((rng.standard_normal((co, ci, k, k)) * np.sqrt(2.0 / (ci * k * k))).astype(np.float32))
        sd[f"features.{idx}.bias"] = torch.from_numpy((0.05 * rng.standard_normal(co)).astype(np.float32))
    for i, c in enumerate(CHANNELS):
        sd[f"lin{i}.model.1.weight"] = torch.from_numpy(np.abs(rng.standard_normal((1, c, 1, 1))).astype(np.float32) / c)
    return sd


def features(sd, x):
    """The five ReLU taps of AlexNet.features on the scaled input, x: (N, 3, H, W)."""
    shift = torch.tensor(SHIFT, dtype=x.dtype).view(1, 3, 1, 1)
    scale = torch.tensor(SCALE, dtype=x.dtype).view(1, 3, 1, 1)
    h = (x - shift) / scale
    taps = []
    for n, (idx, _, _, _, stride, pad) in enumerate(CONVS):
        if n in (1, 2):
            h = F.max_pool2d(h, kernel_size=3, stride=2)
        h = F.relu(F.conv2d(h, sd[f"features.{idx}.weight"].to(x.dtype), sd[f"features.{idx}.bias"].to(x.dtype), stride=stride, padding=pad))
        taps.append(h)
    return taps


def distance(sd, x0, x1):
    """lpips.LPIPS(net='alex').forward(x0, x1) -> (N,) distances."""
    t0, t1 = features(sd, x0), features(sd, x1)
    total = torch.zeros(x0.shape[0], dtype=x0.dtype)
    for i, (a, b) in enumerate(zip(t0, t1)):
        na = a / (a.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        nb = b / (b.pow(2).sum(1, keepdim=True).sqrt() + 1e-10)
        d = (na - nb).pow(2)
        total = total + F.conv2d(d, sd[f"lin{i}.model.1.weight"].to(x0.dtype)).mean((1, 2, 3))
    return total
