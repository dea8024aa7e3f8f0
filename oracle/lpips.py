"""Oracle: LPIPS v0.1 with the AlexNet backbone (test infrastructure only).

The sender's decision rule is built on ``lpips.LPIPS(net='alex')`` (reference ``city_sender.py:302``, called per frame by
``decide_5to5_lpips``, ``:376-406``, on the [0, 1] frames as they are -- no ``normalize=True``).  The pip package
(``lpips==0.1.4``, ``requirements.txt:66``) is not importable here, but the reference vendors the same algorithm and its
trained linear layers, and this file follows that text:

* ``models/networks_basic.py:62-84`` ``PNetLin.forward`` (version '0.1', lpips=True, spatial=False): scale the inputs, take the
  five feature taps, ``normalize_tensor``, squared difference, ``lins[k].model`` (a bias-free 1x1 convolution to one channel;
  its Dropout is inactive in eval), ``spatial_average`` (``:14-15``), sum over the taps;
* ``models/networks_basic.py:86-93`` ``ScalingLayer``: ``(x - shift) / scale`` with shift (-.030, -.088, -.188), scale (.458, .448, .450);
* ``models/eval_models.py:35-37`` ``normalize_tensor``: ``x / (sqrt(sum_c x^2) + 1e-10)``;
* ``models/pretrained_networks.py:56-94`` ``alexnet``: slices [0:2], [2:5], [5:8], [8:10], [10:12] of
  ``torchvision.models.alexnet().features`` = conv 11x11 s4 p2 (3->64), ReLU | MaxPool 3 s2, conv 5x5 p2 (64->192), ReLU |
  MaxPool 3 s2, conv 3x3 p1 (192->384), ReLU | conv 3x3 p1 (384->256), ReLU | conv 3x3 p1 (256->256), ReLU.

Pinned: the linear layers -- ``weights/v0.1/alex.pth`` of the reference, committed as ``tests/golden/lpips_alex_lin.npz``.
UNPINNED: the backbone weights (torchvision's pretrained AlexNet is not in the reference tree and cannot be fetched) and,
since neither torchvision nor skimage is importable, the vendored module cannot be run to produce an output fixture; tests
use seeded stand-in convolutions under the real linear layers.

State-dict key names follow the packages: ``features.{0,3,6,8,10}.{weight,bias}`` (torchvision AlexNet) and
``lin{0..4}.model.1.weight`` of shape (1, C, 1, 1) (``weights/v0.1/alex.pth``).
"""
import numpy as np
import torch
import torch.nn.functional as F

SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)
CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1), (10, 256, 256, 3, 1, 1))
CHANNELS = (64, 192, 384, 256, 256)


def seeded_state_dict(seed, lin=None):
    """Weights in the packages' layouts: He-scaled normal stand-ins for the AlexNet convolutions (the trained ones cannot be
    fetched offline), small biases; ``lin``: the reference's trained linear layers (tests/golden/lpips_alex_lin.npz, keys
    lin0..lin4) -- without it non-negative stand-ins as the trained ones are."""
    rng = np.random.default_rng(seed)
    sd = {}
    for idx, ci, co, k, _, _ in CONVS:
        sd[f"features.{idx}.weight"] = torch.from_numpy((rng.standard_normal((co, ci, k, k)) * np.sqrt(2.0 / (ci * k * k))).astype(np.float32))
        sd[f"features.{idx}.bias"] = torch.from_numpy((0.05 * rng.standard_normal(co)).astype(np.float32))
    for i, c in enumerate(CHANNELS):
        w = np.abs(rng.standard_normal((1, c, 1, 1))).astype(np.float32) / c       # drawn either way: same convolutions per seed
        if lin is not None:
            w = np.asarray(lin[f"lin{i}"], dtype=np.float32).reshape(1, c, 1, 1)
        sd[f"lin{i}.model.1.weight"] = torch.from_numpy(w.copy())
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
