"""CPU tests of the ELIC oracle (parity unpinned, see oracle/__init__.py): what can be asserted without the
reference -- encode/decode symbol identity, decompress(compress(x)) == dense quantised forward, bit counts --
and that the product's synthetic checkpoint matches the oracle's seeded-weight recipe."""
import numpy as np
import torch

import evc_amd  # noqa: F401
from evc_amd import config as C
from evc_amd import lib, synthetic
from oracle import elic as OEL
from oracle import rans as OR
from oracle.scorenet import Dims, seeded_params


class NativeCoder:
    """The C++ coder behind the oracle's coder interface (lets the oracle run at full size quickly)."""
    @staticmethod
    def encode_with_indexes(sym, idx, cdf, size, off):
        return lib.rans_encode(sym, idx, cdf, size, off)

    @staticmethod
    def decode_with_indexes(s, idx, cdf, size, off):
        return lib.rans_decode(s, idx, cdf, size, off)


def test_oracle_elic_roundtrip_python_coder_small():
    p = synthetic.elic_state_dict(7)
    x = torch.from_numpy(synthetic.make_clips(1, seed=1, frames=1, size=64)[0, 0].astype(np.float32) / 255)[None]
    enc = OEL.compress(p, x, coder=OR)
    assert tuple(enc["shape"]) == (1, 1) and len(enc["strings"][0]) == 5 and len(enc["strings"][1]) == 1
    dec = OEL.decompress(p, enc["strings"], enc["shape"], coder=OR, return_latents=True)
    assert torch.equal(dec["y_hat"], enc["y_hat"])          # decoder reproduces the encoder's quantised latents
    assert dec["x_hat"].shape == (1, 3, 64, 64) and float(dec["x_hat"].min()) >= 0 and float(dec["x_hat"].max()) <= 1
    # native and python coders emit the same bytes for the whole frame
    enc2 = OEL.compress(p, x, coder=NativeCoder)
    assert enc2["strings"][1] == enc["strings"][1] and enc2["strings"][0] == enc["strings"][0]
    # residual of the quantised latent is a half-integer-bounded rounding error at every coded site
    assert float((enc["y_hat"] - enc["y"]).abs().max()) <= 0.5 + 1e-4
    assert OEL.count_bits(enc["strings"]) == 8 * (sum(len(s) for sl in enc["strings"][0] for pp in sl for s in pp)
                                                  + sum(len(s) for s in enc["strings"][1]))


def test_oracle_inference_wrapper_pads_and_crops():
    p = synthetic.elic_state_dict(8)
    x = torch.from_numpy(synthetic.make_clips(1, seed=2, frames=1, size=64)[0, 0].astype(np.float32) / 255)
    x = x[:, :40, :56]     # not a multiple of the patch: padded right/bottom, cropped after decode (Inference.py:21-45)
    x_hat, bits = OEL.inference(p, x, patch=64, coder=NativeCoder)
    assert x_hat.shape == (1, 3, 40, 56) and bits > 0 and bits % 32 == 0


def test_synthetic_diffusion_weights_equal_oracle_recipe():
    cfg = C.default_config(32, 32, 32)
    sd = synthetic.diffusion_state_dict(cfg, 11)
    ref = seeded_params(Dims(ngf=32, n_head_channels=32, image_size=32), 11)
    assert list(sd) == list(ref)
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k


def test_exact_backend_restates_the_convolutions_and_round_trips():
    """oracle/exact.py (the bit-exact restatement of the product's fp32 convolution arithmetic, C helper built by
    oracle/Makefile): its convolutions equal torch's within fp32 round-off -- 3x3 / 5x5 / 1x1 "same" convolutions, the masked
    context convolution, and the polyphase form of compressai's deconv (ConvTranspose2d k 5, stride 2, padding 2,
    output_padding 1) -- and the oracle codec run in exact mode round-trips its own streams symbol for symbol."""
    import torch.nn.functional as F
    from oracle import exact as X
    g = torch.Generator().manual_seed(7)
    for (ci, co, k) in ((64, 96, 3), (48, 32, 5), (224, 128, 1), (20, 24, 3)):       # 20: channel padding to 32
        x = torch.randn(2, ci, 8, 8, generator=g)
        w = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
        b = torch.randn(co, generator=g)
        ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=k // 2))
        out = X.conv2d(x, w, b, relu_out=True)
        assert float((out.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max() + 1)
        assert torch.equal(out, X.conv2d(x, w, b, relu_out=True))                      # deterministic
    x = torch.randn(2, 192, 2, 2, generator=g)
    wt = torch.randn(192, 100, 5, 5, generator=g) / (192 * 25) ** 0.5
    b = torch.randn(100, generator=g)
    ref = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=2, output_padding=1)
    out = X.deconv5x5s2(x, wt, b)
    assert out.shape == (2, 100, 4, 4) and float((out.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max() + 1)
    # the codec in exact mode: encode -> decode returns the encoder's latents bit for bit
    p = synthetic.elic_state_dict(5)
    img = torch.rand(1, 3, 64, 64, generator=g)
    enc = OEL.compress(p, img, coder=NativeCoder, exact=True)
    dec = OEL.decompress(p, enc["strings"], enc["shape"], coder=NativeCoder, return_latents=True, exact=True)
    assert torch.equal(dec["y_hat"], enc["y_hat"])
