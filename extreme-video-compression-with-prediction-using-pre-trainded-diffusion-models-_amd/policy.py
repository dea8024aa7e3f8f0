"""Sender policy: which frames are transmitted as ELIC key frames and which are left to the generator.

Restates ``SenderCity.update`` / ``decide_5to5`` / ``decide_5to5_lpips`` and the sweep of the reference's main loop
(city_sender.py:353-437, 495-607): per (video, q, threshold) job, starting from two key frames, generate 5 frames
from the last two decoded ones, keep the longest prefix whose quality metric passes the threshold, and when not a
single frame passes, key-code the next two frames instead.  The reference runs the jobs strictly one after another
(1 video x 2 q x 28 thresholds, a fresh sampler + checkpoint reload per job); they are independent, so here ALL
active jobs -- every (video, q, threshold) -- advance in lockstep: each round stacks their conditioning frames along
the batch axis of one score-network launch (``max_batch`` per launch), and the key frames a round needs are coded in
one batched ELIC call per q, once per (video, q, frame) however many thresholds fall back to it.

Noise: every job draws from its own counter-based stream (seed, job id, round, step), so a job's frames do not depend
on which other jobs share its launch -- the batched sweep and a one-job-at-a-time run make the same decisions.

Metrics: ``PsnrMetric`` is the reference's ``decide_5to5`` rule (accept while PSNR >= threshold, cal_psnr of
city_sender.py:257-260).  ``CallableMetric`` is the hook for ``decide_5to5_lpips`` (accept while distance <=
threshold): LPIPS needs torchvision's pretrained AlexNet, which cannot be fetched offline, so the metric is supplied
by the user (``--policy lpips --metric pkg.module:function`` or the ``lpips`` package when it is importable).
"""
import importlib
import os

import numpy as np
import torch

from .elic import count_bits


def cal_psnr(a, b, maxvalue=1.0):
    """city_sender.py:257-260 (float64)."""
    mse = np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)
    return 10 * np.log10((maxvalue ** 2) / mse)


class PsnrMetric:
    """decide_5to5 (city_sender.py:353-374): a generated frame is kept while PSNR(pred, gt) >= threshold."""
    name = "psnr"

    def values(self, pred, gt):
        p, g = pred.detach().cpu().numpy(), gt.detach().cpu().numpy()
        return np.asarray([cal_psnr(p[i], g[i]) for i in range(p.shape[0])])

    @staticmethod
    def accept(value, thr):
        return value >= thr


class CallableMetric:
    """decide_5to5_lpips (city_sender.py:376-406): kept while distance(pred, gt) <= threshold.  ``fn(pred, gt)`` takes
    two (n, 3, H, W) float tensors in [0, 1] on the device and returns n distances."""
    name = "lpips"

    def __init__(self, fn, name="lpips"):
        self.fn, self.name = fn, name

    def values(self, pred, gt):
        v = self.fn(pred.float(), gt.float())
        return np.asarray(v.detach().cpu().reshape(-1).tolist() if torch.is_tensor(v) else v, dtype=np.float64)

    @staticmethod
    def accept(value, thr):
        return value <= thr


def load_metric(policy, spec=None, device="cuda"):
    """``psnr`` -> PsnrMetric; ``lpips`` -> the HIP LPIPS-AlexNet (``lpips.LpipsAlex``) when ``spec`` names its weight files
    ("alexnet.pth,alex.pth" or one saved LPIPS state dict), else a CallableMetric around ``spec`` = "package.module:callable"
    (called as fn(pred, gt)), or around the ``lpips`` package (LPIPS(net='alex'), inputs scaled to [-1, 1] as lpips expects --
    the reference feeds [0, 1] frames unnormalised, city_sender.py:389-390; pass your own callable to reproduce that)."""
    if policy == "psnr":
        return PsnrMetric()
    if policy != "lpips":
        raise ValueError(f"unknown policy metric {policy!r}")
    if spec and (spec.startswith("file:") or all(os.path.isfile(part) for part in spec.split(","))):
        spec = spec[5:] if spec.startswith("file:") else spec
        # weight files: "alexnet-owt-*.pth,alex.pth" (torchvision backbone + lpips v0.1 linear layers) or one file holding
        # a saved lpips.LPIPS state dict -> the HIP implementation (lpips.py); frames are passed as they are, like the reference
        from .lpips import LpipsAlex
        paths = spec.split(",")
        net = (LpipsAlex.from_files(paths[0], paths[1], device) if len(paths) == 2 else
               LpipsAlex(torch.load(paths[0], map_location="cpu", weights_only=True), device))
        return CallableMetric(lambda a, b: net(a, b), name="lpips-alex-hip")
    if spec:
        mod, _, attr = spec.partition(":")
        fn = getattr(importlib.import_module(mod), attr or "metric")
        return CallableMetric(fn, name=spec)
    try:
        import lpips  # noqa: F401  (needs torchvision's pretrained AlexNet: not available offline)
    except Exception as e:
        raise RuntimeError("--policy lpips needs a perceptual metric: the `lpips` package (with torchvision's pretrained "
                           "AlexNet) is not importable here; pass --metric package.module:callable "
                           "(fn(pred, gt) -> distances for (n,3,H,W) tensors in [0,1]) or use --policy psnr") from e
    net = lpips.LPIPS(net="alex").to(device).eval()
    return CallableMetric(lambda a, b: net(a, b).reshape(-1), name="lpips-alex")


def inference_batch(model, x, patch):
    """Inference.inference (Inference.py:19-75) for a batch: x (n, 3, H, W) in [0, 1] -> (x_hat (n, 3, H, W), bits[n])."""
    n, _, h, w = x.shape
    ph, pw = (-h) % patch, (-w) % patch
    xp = torch.nn.functional.pad(x, (0, pw, 0, ph))
    enc = model.compress(xp)
    dec = model.decompress(enc["strings"], enc["shape"])["x_hat"][:, :, :h, :w]
    ys, zs = enc["strings"]
    bits = [count_bits([[[[p[b]] for p in sl] for sl in ys], [zs[b]]]) for b in range(n)]
    return dec, bits


class _Job:
    __slots__ = ("uid", "vid", "q", "thr", "x", "d", "bits", "round")

    def __init__(self, uid, vid, q, thr):
        self.uid, self.vid, self.q, self.thr = uid, vid, q, thr
        self.x, self.d, self.bits, self.round = [], [], [], 0


def run_policy(decoder, models, clips, qs, thresholds, metric, patch=64, frames=30, max_batch=32, seed=0,
               bpp_limit=1.0, device="cuda", log=None, noise_source=None, stats=None):
    """The reference's sweep, batched.

    decoder:       ClipDecoder (only ``generate`` is used: the generator does not depend on q)
    models:        q -> ElicModel
    clips:         {vid: float tensor (frames, 3, H, W) in [0, 1]} (host)
    noise_source:  optional ``fn(job, round, step, shape) -> tensor`` (job = (vid, q, thr); step 0 = x_T, step i+1 = the
                   noise of sampler step i) replacing the per-job device generators -- parity tests inject the noise the
                   CPU oracle loop uses
    stats:         optional dict, filled with the launch-size histogram {batch size: generation launches}, the number of
                   generation rounds and of key frames coded
    Returns {(vid, q): [dict(thr, x (frames,3,H,W) float32 numpy, d (frames,) int, bits [..], bpp)]} with, per (vid, q),
    the thresholds in the given order cut at the first one whose rate reaches ``bpp_limit`` bits per pixel
    (``if NN_bpp >= 1.0: break``, city_sender.py:563-564)."""
    jobs, uid = [], 0
    for vid in clips:
        for q in qs:
            for thr in thresholds:
                jobs.append(_Job(uid, vid, q, thr))
                uid += 1
    gt_dev = {vid: c.to(device=device, dtype=torch.float32) for vid, c in clips.items()}
    H, W = next(iter(clips.values())).shape[-2:]
    key_cache = {}                  # (vid, q, frame) -> (x_hat on device, bits)

    def ensure_keys(wanted):
        """Code every missing (vid, q, frame) of ``wanted``: one batched ELIC encode + decode per q."""
        for q in qs:
            todo = sorted({k for k in wanted if k[1] == q and k not in key_cache})
            for c0 in range(0, len(todo), max_batch):
                part = todo[c0:c0 + max_batch]
                x = torch.stack([gt_dev[v][f] for (v, _, f) in part], 0)
                xh, bits = inference_batch(models[q], x, patch)
                for i, k in enumerate(part):
                    key_cache[k] = (xh[i], bits[i])

    def add_keys(job, fs):
        for f in fs:
            xh, b = key_cache[(job.vid, job.q, f)]
            job.x.append(xh); job.bits.append(b); job.d.append(1)

    ensure_keys({(j.vid, j.q, f) for j in jobs for f in (0, 1)})          # city_sender.py:521-524
    for j in jobs:
        add_keys(j, (0, 1))

    def noise_for(batch):
        def fn(tag, shape):      # one counter-based stream per (job, round, step): independent of the batch composition
            step = 0 if tag == "init" else int(tag) + 1
            out = torch.empty(shape, device=device, dtype=torch.float32)
            if noise_source is not None:
                for i, j in enumerate(batch):
                    out[i] = noise_source((j.vid, j.q, j.thr), j.round, step, tuple(shape[1:])).to(device)
                return out
            for i, j in enumerate(batch):
                g = torch.Generator(device=device)
                g.manual_seed(((((int(seed) & 0xFFFFF) << 20 | j.uid) << 6 | j.round) << 10 | step) & (2 ** 63 - 1))
                out[i] = torch.randn(shape[1:], device=device, dtype=torch.float32, generator=g)
            return out
        return fn

    while True:
        active = [j for j in jobs if len(j.x) < frames]
        if not active:
            break
        fallback = []
        for c0 in range(0, len(active), max_batch):
            batch = active[c0:c0 + max_batch]
            cond = torch.stack([torch.stack(j.x[-2:], 0) for j in batch], 0).contiguous()     # (n, 2, 3, H, W)
            pred = decoder.generate(cond, noise_fn=noise_for(batch), groups=1)              # (n, 5, 3, H, W)
            if stats is not None:
                h = stats.setdefault("launch_sizes", {})
                h[len(batch)] = h.get(len(batch), 0) + 1
            n_new = [min(pred.shape[1], frames - len(j.x)) for j in batch]
            flat_p = torch.cat([pred[k, :n_new[k]] for k in range(len(batch))], 0)
            flat_g = torch.cat([gt_dev[j.vid][len(j.x):len(j.x) + n_new[k]] for k, j in enumerate(batch)], 0)
            vals = metric.values(flat_p, flat_g)                 # every candidate frame of the launch in one call
            o = 0
            for k, j in enumerate(batch):
                acc = 0
                for t in range(n_new[k]):
                    if not metric.accept(vals[o + t], j.thr):
                        break
                    j.x.append(pred[k, t]); j.d.append(0); acc += 1
                o += n_new[k]
                j.round += 1
                if acc == 0:
                    fallback.append(j)
        if fallback:                 # city_sender.py:538-548: key-code the next two frames
            ensure_keys({(j.vid, j.q, f) for j in fallback for f in (len(j.x), len(j.x) + 1) if f < frames})
            for j in fallback:
                add_keys(j, [f for f in (len(j.x), len(j.x) + 1) if f < frames])
        if stats is not None:
            stats["rounds"] = stats.get("rounds", 0) + 1
            stats["key_frames_coded"] = len(key_cache)
        if log is not None:
            log(f"policy round: {len(active)} active jobs, {len(fallback)} fell back to key frames")

    out = {}
    for j in jobs:
        lst = out.setdefault((j.vid, j.q), [])
        if lst and lst[-1] is None:
            continue                                            # sweep of this (vid, q) already cut
        bpp = sum(j.bits) / H / W / frames
        if bpp >= bpp_limit:
            lst.append(None)
            continue
        lst.append(dict(thr=j.thr, x=torch.stack(j.x[:frames], 0).cpu().numpy(), d=np.asarray(j.d[:frames], dtype=np.int64),
                        bits=list(j.bits), bpp=bpp))
    return {k: [r for r in v if r is not None] for k, v in out.items()}


def rd_envelope(bpp, metric_mean, higher_is_better):
    """The rate-distortion frontier of one video's sweep: the part of the convex hull of the (bpp, metric) points
    between the leftmost point and the best-metric point -- upper-left chain for PSNR, lower-left chain for LPIPS --
    which is what ``process_data_and_save`` keeps (function.py:148-204).  scipy's ConvexHull lists the vertices
    counter-clockwise from an arbitrary start; the reference slices that list with index arithmetic that is only
    right when the start happens to fall outside the chain (``range(highest, leftmost + 1)``), so the chain is walked
    cyclically here instead.  Returns a (2, n) array [bpp; metric]; fewer than 3 points (or collinear ones) are
    returned as they are."""
    pts = np.stack([np.asarray(bpp, dtype=np.float64), np.asarray(metric_mean, dtype=np.float64)], 1)
    if len(pts) < 3:
        return pts.T.copy()
    try:
        from scipy.spatial import ConvexHull
        hull = ConvexHull(points=pts)
    except Exception:
        return pts.T.copy()
    v = list(hull.vertices)
    n = len(v)
    left = int(np.argmin(pts[v, 0]))
    if higher_is_better:       # counter-clockwise, the upper chain runs from the highest point back to the leftmost
        start, stop = int(np.argmax(pts[v, 1])), left
    else:                      # ... and the lower chain from the leftmost point on to the lowest
        start, stop = left, int(np.argmin(pts[v, 1]))
    sel, i = [start], start
    while i != stop:
        i = (i + 1) % n
        sel.append(i)
    return pts[[v[i] for i in sel]].T.copy()
