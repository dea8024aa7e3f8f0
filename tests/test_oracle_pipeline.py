"""BASELINE.json configs[0] -- "city_bonn.npy start_idx=0 end_idx=1, q3, PyTorch-CPU reference path (plumbing, no GPU)":
the whole sender/receiver loop of the reference on the CPU oracle, two clips, reduced sizes so it runs in seconds
(synthetic clips and seeded weights: the dataset and checkpoints are not available offline).  Every stage the loop
strings together is pinned elsewhere (score network + samplers by the reference's goldens, the range coder against the
native one); the GPU path's per-stage parity against these same oracle functions is in tests/test_gpu_*.py, and the GPU
receiver against this loop's generation step in tests/test_gpu_decoder.py."""
import numpy as np
import torch

from conftest import rnd
from oracle import pipeline as OP
from oracle import scorenet as ON


class NativeCoder:
    """compressai's coder is C++; the oracle's pure-Python one is bit-identical (tests/test_rans_entropy.py) but slow."""
    @staticmethod
    def encode_with_indexes(sym, idx, cdf, size, off):
        from evc_amd import lib
        return lib.rans_encode(sym, idx, cdf, size, off)

    @staticmethod
    def decode_with_indexes(s, idx, cdf, size, off):
        from evc_amd import lib
        return lib.rans_decode(s, idx, cdf, size, off)


def test_configs0_two_clip_plumbing_run_on_the_cpu_oracle():
    import evc_amd  # noqa: F401
    from evc_amd import synthetic
    size, frames, subsample = 64, 12, 2
    d_net = ON.Dims(ngf=32, n_head_channels=32, image_size=size)
    p_net = ON.seeded_params(d_net, 9)
    p_elic = synthetic.elic_state_dict(3)                                        # q3
    clips = synthetic.make_clips(2, seed=0, frames=frames, size=size).astype(np.float64) / 255.0   # start 0, end 1

    def noise_fn(tag, shape):
        return rnd(hash(str(tag)) % 10007, *shape)
    results = []
    for vid in (0, 1):                                                           # city_sender.py:495: end_idx inclusive
        gt = torch.from_numpy(clips[vid])
        r_all = OP.run_clip(p_net, d_net, p_elic, gt, threshold=-100.0, subsample=subsample, noise_fn=noise_fn,
                            coder=NativeCoder, frames=frames)
        r_none = OP.run_clip(p_net, d_net, p_elic, gt, threshold=200.0, subsample=subsample, noise_fn=noise_fn,
                             coder=NativeCoder, frames=frames)
        results.append((r_all, r_none))
        for r in (r_all, r_none):
            assert r["x"].shape == (frames, 3, size, size) and float(r["x"].min()) >= 0 and float(r["x"].max()) <= 1
            assert len(r["d"]) == frames and r["d"][0] == r["d"][1] == 1
            assert len(r["bits"]) == int(r["d"].sum()) and all(b > 0 and b % 8 == 0 for b in r["bits"])
            assert abs(r["bpp"] - sum(r["bits"]) / size / size / frames) < 1e-12
            assert len(r["psnr"]) == frames and np.isfinite(r["psnr"]).all()
        assert r_all["d"].sum() == 2                   # every generated frame accepted: only the initial key frames
        assert r_none["d"].sum() == frames             # nothing accepted: all frames key-coded, pairs at a time
        assert r_none["bpp"] > 3 * r_all["bpp"]
        # (no quality assertion: with seeded random weights neither codec output resembles the ground truth)
        # the two jobs share their first two (key) frames bit for bit
        assert torch.equal(r_all["x"][:2], r_none["x"][:2])
    # determinism of the whole loop (same seeded noise): a second run reproduces the first
    gt0 = torch.from_numpy(clips[0])
    again = OP.run_clip(p_net, d_net, p_elic, gt0, threshold=-100.0, subsample=subsample, noise_fn=noise_fn,
                        coder=NativeCoder, frames=frames)
    assert torch.equal(again["x"], results[0][0]["x"]) and again["bits"] == results[0][0]["bits"]
