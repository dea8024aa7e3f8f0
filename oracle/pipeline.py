"""Oracle: the reference's whole sender/receiver loop for one clip on the CPU -- test infrastructure only.

Restates the main loop of ``city_sender.py:495-607`` with ``SenderCity.update`` / ``decide_5to5`` (:353-437) and
``compress`` (:440-461) on top of the oracle's stages (oracle/elic.py, oracle/scorenet.py, oracle/samplers.py): two key
frames coded with ELIC, then chunks of 5 generated frames conditioned on the last two decoded ones, accepted while their
PSNR passes the threshold, and two more key frames whenever a chunk yields nothing.  BASELINE.json configs[0]
("start_idx=0 end_idx=1, q3, PyTorch-CPU reference path, plumbing, no GPU") is this loop on 2 clips.
"""
import numpy as np
import torch

from . import elic as OE
from . import samplers as OS
from . import schedule as OSch
from . import scorenet as ON


def cal_psnr(a, b, maxvalue=1.0):
    """city_sender.py:257-260."""
    mse = np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)
    return 10 * np.log10((maxvalue ** 2) / mse)


def generate_chunk(p_net, d_net, prev2, subsample, noise_fn, chunk_id):
    """SenderCity.generate_frame (city_sender.py:326-351): prev2 (2,3,H,W) in [0,1] -> (5,3,H,W) in [0,1]."""
    _, C, H, W = prev2.shape
    cond = (2.0 * prev2.reshape(1, 2 * C, H, W) - 1.0).float()                  # data_transform
    x_T = noise_fn((chunk_id, "init"), (1, 5 * C, H, W))
    x = OS.ddpm(x_T.clone(), lambda xx, t: ON.forward(p_net, d_net, xx, t, cond=cond), OSch.base_schedule(),
                subsample_steps=subsample, noise_fn=lambda i, xx: noise_fn((chunk_id, i), tuple(xx.shape)))
    return ((x[0] + 1.0) / 2.0).clamp(0.0, 1.0).reshape(5, C, H, W)             # inverse_data_transform


def run_clip(p_net, d_net, p_elic, gt, threshold, subsample, noise_fn, patch=64, coder=None, frames=30, trace=None,
             distance=None):
    """One (video, q, threshold) job of the reference's sweep.  gt: (frames,3,H,W) float in [0,1].
    Returns dict(x (frames,3,H,W), d (frames,), bits [per key frame], bpp, psnr [per frame]).  ``trace``: optional list
    that receives every value the accept / reject rule looked at (tests use it to keep thresholds away from ties).
    ``distance``: None = decide_5to5 (PSNR, kept while >= threshold, city_sender.py:353-374); a callable
    (pred_frame, gt_frame) -> float = decide_5to5_lpips (kept while <= threshold, :376-406)."""
    kw = {} if coder is None else {"coder": coder}
    x, d, bits = [], [], []

    def key(f):
        xh, b = OE.inference(p_elic, gt[f].float(), patch=patch, **kw)
        x.append(xh[0]); d.append(1); bits.append(b)
    key(0); key(1)                                                                  # city_sender.py:521-524
    chunk = 0
    while len(x) < frames:
        l = len(x)
        pred = generate_chunk(p_net, d_net, torch.stack(x[-2:], 0), subsample, noise_fn, chunk)
        chunk += 1
        acc = 0
        for t in range(min(5, frames - l)):                                         # decide_5to5: accepted prefix
            v = cal_psnr(pred[t].numpy(), gt[l + t].numpy()) if distance is None else float(distance(pred[t], gt[l + t]))
            if trace is not None:
                trace.append(v)
            if (v < threshold) if distance is None else (v > threshold):
                break
            x.append(pred[t]); d.append(0); acc += 1
        if acc == 0:                                                                # :538-548: two more key frames
            for f in (l, l + 1):
                if f < frames:
                    key(f)
    xs = torch.stack(x[:frames], 0)
    H, W = gt.shape[-2:]
    return dict(x=xs, d=np.asarray(d[:frames]), bits=bits, bpp=sum(bits) / H / W / frames,
                psnr=[cal_psnr(xs[i].numpy(), gt[i].numpy()) for i in range(frames)])
