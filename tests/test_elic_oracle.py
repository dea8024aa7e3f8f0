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
