"""LPIPS (AlexNet, v0.1) on the HIP kernels: the perceptual distance of the sender's decision rule.

Mirrors ``lpips.LPIPS(net='alex')`` as the reference builds and calls it (``city_sender.py:302``; ``decide_5to5_lpips``,
``:376-406``: one (3, H, W) frame pair at a time, [0, 1] data passed as is).  The algorithm is the one the reference vendors in
``models/networks_basic.py:62-93`` (``PNetLin.forward``, ``ScalingLayer``), ``models/eval_models.py:35-37`` and
``models/pretrained_networks.py:56-94`` (oracle/lpips.py restates it).  Weights: ``LpipsAlex`` takes a state dict in the packages'
own key names -- ``features.{0,3,6,8,10}.{weight,bias}`` (torchvision ``alexnet``; not in the reference tree, cannot be fetched
offline) plus ``lin{k}.model.1.weight`` (the reference's ``weights/v0.1/alex.pth``; the ``lins.{k}.`` and ``net.slice*``
spellings of a saved ``LPIPS`` module are accepted too) -- or the two files themselves (``from_files``).

Data path: NCHW frames -> ``evc_im2col_nchw_f32`` (ScalingLayer + 11x11 stride-4 patches) -> 1x1 convolution, then
``evc_maxpool3s2_nhwc_f32`` and 5x5 / 3x3 convolutions, all on ``evc_conv2d_nhwc_f32`` with the exact bf16 split (the inputs are
not normalised) and ReLU fused in the epilogue; ``evc_lpips_layer_f32`` per tap.  Both images of a pair go through the
backbone as one batch.
"""
import torch

from . import lib as L

SHIFT = (-0.030, -0.088, -0.188)
SCALE = (0.458, 0.448, 0.450)
# (features index, Ci, Co, kernel, stride, pad) -- torchvision.models.alexnet().features
CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1), (10, 256, 256, 3, 1, 1))
# where a saved lpips.LPIPS module keeps the backbone convolutions (lpips/pretrained_networks.py: alexnet slices)
_SLICE_KEYS = {0: "net.slice1.0", 3: "net.slice2.3", 6: "net.slice3.6", 8: "net.slice4.8", 10: "net.slice5.10"}
_IM2COL_LD = 368                                       # 3 * 11 * 11 = 363 patch entries, padded to a multiple of 16


def _find(sd, *names):
    for n in names:
        if n in sd:
            return sd[n]
    raise KeyError(f"LPIPS state dict holds none of {names}")


class LpipsAlex:
    def __init__(self, state_dict, device=None):
        L.hip_lib()                                                  # fails loudly without the HIP library / a gfx950 device
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.arith = L.default_arith()
        dev = self.device
        self.shift = torch.tensor(SHIFT, dtype=torch.float32, device=dev)
        self.scale = torch.tensor(SCALE, dtype=torch.float32, device=dev)
        self.layers = []
        for idx, ci, co, k, stride, pad in CONVS:
            w = _find(state_dict, f"features.{idx}.weight", f"{_SLICE_KEYS[idx]}.weight").detach().float().to(dev)
            b = _find(state_dict, f"features.{idx}.bias", f"{_SLICE_KEYS[idx]}.bias").detach().float().to(dev).contiguous()
            if tuple(w.shape) != (co, ci, k, k) or tuple(b.shape) != (co,):
                raise ValueError(f"features.{idx}: expected weight {(co, ci, k, k)}, got {tuple(w.shape)}")
            if idx == 0:                                             # as a 1x1 convolution over the im2col rows
                wp = torch.zeros((co, _IM2COL_LD, 1, 1), dtype=torch.float32, device=dev)
                wp[:, :ci * k * k, 0, 0] = w.reshape(co, -1)
                w, k = wp, 1
            self.layers.append(dict(w=L.conv_pack_weights(w.contiguous(), self.arith), b=b, co=co, k=k))
        self.lins = []
        for i, (_, _, co, _, _, _) in enumerate(CONVS):
            w = _find(state_dict, f"lin{i}.model.1.weight", f"lins.{i}.model.1.weight").detach().float().to(dev)
            if w.numel() != co:
                raise ValueError(f"lin{i}: expected {co} weights, got {tuple(w.shape)}")
            self.lins.append(w.reshape(-1).contiguous())

    @classmethod
    def from_files(cls, alexnet_path, lin_path, device=None):
        """torchvision's ``alexnet-owt-*.pth`` and lpips' ``weights/v0.1/alex.pth``, read with the weights-only loader."""
        sd = dict(torch.load(alexnet_path, map_location="cpu", weights_only=True))
        sd.update(torch.load(lin_path, map_location="cpu", weights_only=True))
        return cls(sd, device)

    def features(self, x):
        """x: (N, 3, H, W) float32 on the device -> the five ReLU taps, NHWC."""
        h = L.im2col_nchw(x.contiguous(), 11, 11, 4, 2, _IM2COL_LD, self.shift, self.scale)
        taps = []
        for n, e in enumerate(self.layers):
            if n in (1, 2):
                h = L.maxpool3s2_nhwc(h)
            h = L.conv2d_nhwc(h, e["w"], e["co"], e["k"], e["k"], bias=e["b"], act_out=L.ACT_RELU)
            taps.append(h)
        return taps

    def __call__(self, in0, in1):
        """Distances of image pairs, each (3, H, W) or (N, 3, H, W) -> (N,) tensor on the device (lpips returns (N, 1, 1, 1))."""
        a = in0.to(self.device, torch.float32)
        b = in1.to(self.device, torch.float32)
        if a.dim() == 3:
            a, b = a[None], b[None]
        n = a.shape[0]
        taps = self.features(torch.cat([a, b], 0))
        dist = torch.empty(n, dtype=torch.float32, device=self.device)
        for i, t in enumerate(taps):
            L.lpips_layer(t[:n], t[n:], self.lins[i], dist, accumulate=i > 0)
        return dist
