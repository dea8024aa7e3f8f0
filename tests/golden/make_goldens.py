#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own diffusion code on CPU.

Runs ONLY in the build container, where the reference is mounted read-only at /root/reference
(it does not exist on the GPU box).  Nothing from the reference is copied: this script imports
it, feeds it seeded inputs / seeded weights and stores inputs+outputs as small ``.npz`` fixtures
next to this file.  Tests never import the reference; they rebuild the same seeded inputs and
compare against the stored outputs.

    python tests/golden/make_goldens.py            # all fixtures
    python tests/golden/make_goldens.py --skip-full  # skip the full-size (262 M param) forward

The ELIC path (Network.py / ELICUtilis) is NOT covered: it needs compressai/timm/thop/ptflops,
none of which is installed here (ordinary ModuleNotFoundError) -- see oracle/__init__.py.
"""
import argparse
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

from oracle.scorenet import Dims, seeded_params  # noqa: E402  (seed recipe shared with the tests)


def d2n(d):
    ns = argparse.Namespace()
    for k, v in d.items():
        setattr(ns, k, d2n(v) if isinstance(v, dict) else v)
    return ns


def ref_config(ngf, head, image_size, gamma=False, **model_flags):
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/mine.yml")))
    cfg["model"]["gamma"] = gamma
    cfg["model"].update(model_flags)
    cfg["model"]["ngf"] = ngf
    cfg["model"]["n_head_channels"] = head
    cfg["data"]["image_size"] = image_size
    config = d2n(cfg)
    config.device = torch.device("cpu")
    return config


def ref_net(ngf, head, image_size, seed, gamma=False, **model_flags):
    from models.better.ncsnpp_more import UNetMore_DDPM
    net = UNetMore_DDPM(ref_config(ngf, head, image_size, gamma, **model_flags)).eval()
    d = Dims(ngf=ngf, n_head_channels=head, image_size=image_size, cond_emb=bool(model_flags.get("cond_emb", False)))
    p = seeded_params(d, seed)
    own = dict(net.named_parameters())
    assert set(own) == set(p), (sorted(set(own) ^ set(p))[:8])
    for k, v in p.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    missing, unexpected = net.load_state_dict(p, strict=False)
    assert not unexpected and all(m in ("betas", "alphas", "alphas_prev", "unet.sigmas", "k", "k_cum", "theta_t")
                                  for m in missing), missing
    return net, d


def rnd(seed, *shape):
    return torch.from_numpy(np.random.default_rng(seed).standard_normal(shape, dtype=np.float32))


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_schedule():
    net, _ = ref_net(32, 32, 32, 1)
    out = dict(betas=net.betas, alphas=net.alphas, alphas_prev=net.alphas_prev)
    # the subsampled tables as the reference sampler derives them (models/__init__.py:231-239)
    for S in (10, 50, 100):
        skip = 1000 // S
        steps = torch.tensor(list(range(0, 1000, skip)))
        a = net.alphas.index_select(0, steps)
        ap = torch.cat([a[1:], torch.tensor([1.0]).to(a)])
        b = 1.0 - torch.div(a, ap)
        out.update({f"steps_{S}": steps, f"alphas_{S}": a, f"alphas_prev_{S}": ap, f"betas_{S}": b})
    save("schedule", **out)


def gen_fir():
    from models.better.up_or_down_sampling import upsample_2d, downsample_2d
    x = rnd(11, 2, 6, 16, 16)
    save("fir", x=x, up=upsample_2d(x, [1, 3, 3, 1], factor=2), down=downsample_2d(x, [1, 3, 3, 1], factor=2))
    from models.better.op.upfirdn2d import upfirdn2d
    x2 = rnd(12, 1, 3, 9, 7)
    k = rnd(13, 3, 3)
    save("upfirdn2d_generic", x=x2, k=k,
         up2_pad10=upfirdn2d(x2, k, up=2, down=1, pad=(1, 0)),
         down3_pad21=upfirdn2d(x2, k, up=1, down=3, pad=(2, 1)),
         up3_down2_pad22=upfirdn2d(x2, k, up=3, down=2, pad=(2, 2)))


def gen_blocks():
    """Per-module outputs of a mid-sized net (ngf=64, head 64, 32x32) via forward hooks."""
    net, d = ref_net(64, 64, 32, 21)
    x, cond = rnd(22, 2, 15, 32, 32), rnd(23, 2, 6, 32, 32)
    labels = torch.tensor([430, 430])
    taps = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, idx=idx: taps.__setitem__(idx, o.detach().clone()))
             for idx, m in enumerate(net.unet.all_modules)]
    with torch.no_grad():
        out = net(x, labels, cond=cond)
    for h in hooks:
        h.remove()
    # strided samples of every module's output keep the file small but pin every block
    samples = {}
    for idx, t in taps.items():
        flat = t.reshape(-1)
        stride = max(1, flat.numel() // 512)
        samples[f"tap{idx}"] = flat[::stride][:512].clone()
        samples[f"tapstat{idx}"] = torch.stack([t.mean(), t.std()])
    save("blocks_ngf64", out=out, labels=labels, **samples)


def gen_forward_reduced():
    net, d = ref_net(32, 32, 32, 31)
    x, cond = rnd(32, 2, 15, 32, 32), rnd(33, 2, 6, 32, 32)
    with torch.no_grad():
        o0 = net(x, torch.tensor([0, 0]), cond=cond)
        o1 = net(x, torch.tensor([990, 990]), cond=cond)
        o2 = net(x, torch.tensor([-0.5, -0.5]), cond=cond)  # F-PNDM feeds fractional/negative labels
    save("forward_ngf32", out_t0=o0, out_t990=o1, out_tm05=o2)


def gen_samplers():
    import models
    from models import ddpm_sampler, ddim_sampler, FPNDM_sampler
    net, d = ref_net(32, 32, 32, 41)
    x_T, cond = rnd(42, 2, 15, 32, 32), rnd(43, 2, 6, 32, 32)
    S = 5
    noises = [rnd(100 + i, 2, 15, 32, 32) for i in range(S)]
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: next(it)   # inject the per-step noise the reference would draw
    try:
        ddpm = ddpm_sampler(x_T.clone(), net, cond=cond, subsample_steps=S, denoise=True, clip_before=True,
                            final_only=True, t_min=-1, log=True)
    finally:
        torch.randn_like = orig
    ddim = ddim_sampler(x_T.clone(), net, cond=cond, subsample_steps=S, denoise=True, clip_before=True,
                        final_only=True, t_min=-1, log=True)
    fp = FPNDM_sampler(x_T.clone(), net, cond=cond, subsample_steps=10, final_only=True, clip_before=True)
    save("samplers_ngf32", ddpm=ddpm, ddim=ddim, fpndm=fp)

    # label sequences (incl. the L-1 denoise label quirk and F-PNDM's fractional labels) with a fake net
    class Fake(torch.nn.Module):
        def __init__(self, real):
            super().__init__()
            self.alphas, self.alphas_prev, self.betas = real.alphas, real.alphas_prev, real.betas
            self.type = "v1"
            self.log = []

        def forward(self, x, y, cond=None, cond_mask=None):
            self.log.append(float(y[0]))
            return 0.1 * x
    out = {}
    for nm, fn, kw in (("ddpm", ddpm_sampler, dict(subsample_steps=2, denoise=True)),
                       ("ddim", ddim_sampler, dict(subsample_steps=4, denoise=True)),
                       ("fpndm", FPNDM_sampler, dict(subsample_steps=4)),
                       ("fpndm10", FPNDM_sampler, dict(subsample_steps=10)),
                       ("ddpm100", ddpm_sampler, dict(subsample_steps=100, denoise=True))):
        f = Fake(net)
        fn(x_T.clone(), f, cond=cond, final_only=True, **kw)
        out["labels_" + nm] = np.asarray(f.log, dtype=np.float64)
    save("label_sequences", **out)


def gen_sampler_options():
    """The DDPM / DDIM options the shipped CLI never sets: ``t_min`` > 0 (start from a clean frame, skip the early steps,
    noise the input to the first executed level; models/__init__.py:263-277, :145-157) and ``frac_steps`` (only the last
    fraction of the un-subsampled schedule; :248-256).  Reduced net, injected noise."""
    from models import ddpm_sampler, ddim_sampler
    net, d = ref_net(32, 32, 32, 41)
    x0, cond = rnd(46, 2, 15, 32, 32).clamp(-1, 1), rnd(43, 2, 6, 32, 32)
    out = {}
    orig = torch.randn_like
    for name, fn, kw, n_noise in (("ddpm_tmin", ddpm_sampler, dict(subsample_steps=10, t_min=0.35), 12),
                                  ("ddim_tmin", ddim_sampler, dict(subsample_steps=10, t_min=0.35), 1),
                                  ("ddpm_frac", ddpm_sampler, dict(frac_steps=0.006), 8)):
        noises = [rnd(300 + i, 2, 15, 32, 32) for i in range(n_noise)]
        it = iter(noises)
        torch.randn_like = lambda t, **k: next(it)
        try:
            out[name] = fn(x0.clone(), net, cond=cond, denoise=True, clip_before=True, final_only=True, log=False, **kw)
        finally:
            torch.randn_like = orig
        out[name + "_noises_used"] = np.asarray(n_noise - len(list(it)))
    save("sampler_options", **out)


def gen_sampler_gamma():
    """``gamma=True`` (models/__init__.py:119-153, :225-278, :321-324) on a ``config.model.gamma`` model: the schedule
    buffers k / k_cum / theta_t (ncsnpp_more.py:744-749) and DDPM / DDIM runs whose Gamma draws are injected -- the
    reference's ``Gamma(...).sample`` is replaced by a seeded positive tensor around the distribution's mean
    (k theta + n / 2, n standard normal), recomputable by the tests so both sides consume the identical raw draw."""
    import models as M
    net, d = ref_net(32, 32, 32, 41, gamma=True)
    x0, cond = rnd(46, 2, 15, 32, 32).clamp(-1, 1), rnd(43, 2, 6, 32, 32)
    out = dict(k=net.k, k_cum=net.k_cum, theta_t=net.theta_t)
    real = M.Gamma

    class Injected:
        count = 0

        def __init__(self, conc, rate):
            self.k, self.theta = conc, 1.0 / rate

        def sample(self, shape):
            n = rnd(500 + Injected.count, *shape, *self.k.shape)
            Injected.count += 1
            z = self.k * self.theta + 0.5 * n        # IEEE mul / add only: recomputable bit for bit on any host
            Injected.raws.append(z.clone())
            return z
    for name, fn, kw in (("ddpm_gamma", M.ddpm_sampler, dict(subsample_steps=10)),
                         ("ddpm_gamma_tmin", M.ddpm_sampler, dict(subsample_steps=10, t_min=0.35)),
                         ("ddim_gamma_tmin", M.ddim_sampler, dict(subsample_steps=10, t_min=0.35))):
        Injected.raws = []
        M.Gamma = Injected
        try:
            out[name] = fn(x0.clone(), net, cond=cond, denoise=True, clip_before=True, final_only=True, log=False,
                           gamma=True, **kw)
        finally:
            M.Gamma = real
        # the raw draws are recomputable (``gamma_raw`` below, IEEE fp32 elementwise ops): keep their checksums only
        out[name + "_raw_sums"] = torch.stack([r.double().sum() for r in Injected.raws])
    save("sampler_gamma", **out)


def gamma_raw(count, k, theta, shape):
    """The injected stand-in for draw number ``count`` of ``gen_sampler_gamma``: k theta + n / 2, n = rnd(500 + count),
    in fp32 with k / theta broadcast as full tensors, exactly as ``Injected.sample`` computes it."""
    kk = torch.full(shape[1:], k)
    th = 1.0 / torch.full(shape[1:], 1 / theta)
    return kk * th + 0.5 * rnd(500 + count, *shape)


def gen_model_options():
    """Options of UNetMore_DDPM no shipped config sets (reduced net, ncsnpp_more.py:61,97-99,282-285,735-768): ``cond_emb`` (an
    Embedding(2, ngf/2) row chosen by cond_mask extends the time embedding), ``noise_in_cond`` (the conditioning frames are
    noised to the step's level inside forward; the draw is injected), ``sigma_dist: cosine`` (schedule buffers + a DDPM run)."""
    from models import ddpm_sampler
    x, cond = rnd(61, 2, 15, 32, 32), rnd(62, 2, 6, 32, 32)
    out = {}
    with torch.no_grad():
        net, _ = ref_net(32, 32, 32, 43, cond_emb=True)
        out["cond_emb_default"] = net(x, torch.tensor([500, 7]), cond=cond)
        out["cond_emb_mask10"] = net(x, torch.tensor([500, 7]), cond=cond, cond_mask=torch.tensor([1, 0], dtype=torch.int32))
        net, _ = ref_net(32, 32, 32, 44, noise_in_cond=True)
        z = rnd(63, 2, 6, 32, 32)
        orig = torch.randn_like
        torch.randn_like = lambda t, **k: z
        try:
            out["noise_in_cond"] = net(x, torch.tensor([500, 7]), cond=cond)
        finally:
            torch.randn_like = orig
        net, _ = ref_net(32, 32, 32, 45, sigma_dist="cosine")
        out.update(cos_betas=net.betas, cos_alphas=net.alphas, cos_alphas_prev=net.alphas_prev)
        noises = [rnd(640 + i, 2, 15, 32, 32) for i in range(12)]
        it = iter(noises)
        torch.randn_like = lambda t, **k: next(it)
        try:
            out["cos_ddpm"] = ddpm_sampler(x.clone(), net, cond=cond, denoise=True, clip_before=True, final_only=True, log=False,
                                           subsample_steps=10)
        finally:
            torch.randn_like = orig
        out["cos_noises_used"] = np.asarray(12 - len(list(it)))
    save("model_options", **out)


def gen_lpips_lin():
    """The trained linear layers of LPIPS-AlexNet v0.1 as the reference ships them (weights/v0.1/alex.pth; identical copies under
    models/weights and benchmark/weights), read with the weights-only loader: five non-negative vectors, 1 152 floats.  The
    AlexNet backbone they sit on is torchvision's pretrained checkpoint, which the reference does not hold."""
    sd = torch.load(os.path.join(REF, "weights/v0.1/alex.pth"), map_location="cpu", weights_only=True)
    save("lpips_alex_lin", **{f"lin{i}": sd[f"lin{i}.model.1.weight"].reshape(-1) for i in range(5)})


def gen_forward_full():
    torch.set_num_threads(8)
    net, d = ref_net(192, 192, 128, 1234)
    x, cond = rnd(51, 1, 15, 128, 128), rnd(52, 1, 6, 128, 128)
    with torch.no_grad():
        o = net(x, torch.tensor([500]), cond=cond)
    flat = o.reshape(-1)
    save("forward_full", samples=flat[::60].clone(), stats=torch.stack([o.mean(), o.std(), o.abs().max()]),
         first_row=o[0, :, 0, :].clone())


def gen_forward_full_b9():
    """BASELINE configs[1] launches the score network at B=9 (configs[4] at B=32): nine differently seeded samples
    through the full-size reference in one batch at the DDPM label 500, the first two also at the fractional
    F-PNDM label -0.5 and the denoise label 99.  The GPU tests run them as ONE B=9 launch (and cycled to B=32)."""
    torch.set_num_threads(8)
    net, d = ref_net(192, 192, 128, 1234)
    x = torch.cat([rnd(600 + i, 1, 15, 128, 128) for i in range(9)], 0)
    cond = torch.cat([rnd(700 + i, 1, 6, 128, 128) for i in range(9)], 0)
    with torch.no_grad():
        o = net(x, torch.tensor([500] * 9), cond=cond)
        of = net(x[:2], torch.tensor([-0.5, -0.5]), cond=cond[:2])
        od = net(x[:2], torch.tensor([99, 99]), cond=cond[:2])
    save("forward_full_b9", samples=o.reshape(9, -1)[:, ::60].clone(),
         stats=torch.stack([o.mean((1, 2, 3)), o.std((1, 2, 3)), o.abs().amax((1, 2, 3))], 1),
         samples_tm05=of.reshape(2, -1)[:, ::60].clone(), samples_t99=od.reshape(2, -1)[:, ::60].clone())


def gen_traj_full():
    """Full-size 5-step DDPM trajectory (5 steps + denoise = 6 reference forwards), B=2, injected noise."""
    from models import ddpm_sampler
    torch.set_num_threads(8)
    net, d = ref_net(192, 192, 128, 1234)
    x_T, cond = rnd(801, 2, 15, 128, 128), rnd(802, 2, 6, 128, 128)
    S = 5
    noises = [rnd(810 + i, 2, 15, 128, 128) for i in range(S)]
    it = iter(noises)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: next(it)
    try:
        out = ddpm_sampler(x_T.clone(), net, cond=cond, subsample_steps=S, denoise=True, clip_before=True,
                           final_only=True, t_min=-1, log=True)
    finally:
        torch.randn_like = orig
    o = out[0]
    save("traj_full", samples=o.reshape(2, -1)[:, ::30].clone(), first_row=o[:, :, 0, :].clone(),
         stats=torch.stack([o.mean(), o.std(), o.abs().max()]))


def gen_traj_fpndm_full():
    """BASELINE configs[4]'s sampler at full size: F-PNDM with 10 subsampled steps (3 Runge-Kutta warm-up iterations of 4
    forwards + 7 Adams-Bashforth ones = 19 full-size reference forwards, fractional and negative labels), B=2.
    Reference: models/__init__.py:39-100, models/pndm.py:3-52."""
    from models import FPNDM_sampler
    torch.set_num_threads(8)
    net, d = ref_net(192, 192, 128, 1234)
    x_T, cond = rnd(821, 2, 15, 128, 128), rnd(822, 2, 6, 128, 128)
    out = FPNDM_sampler(x_T.clone(), net, cond=cond, subsample_steps=10, final_only=True, clip_before=True)
    o = out[0] if out.dim() == 5 else out
    save("traj_fpndm_full", samples=o.reshape(2, -1)[:, ::30].clone(), first_row=o[:, :, 0, :].clone(),
         stats=torch.stack([o.mean(), o.std(), o.abs().max()]))


def gen_traj_ddim_full():
    """Full-size 5-step DDIM trajectory (5 deterministic steps + the denoise call = 6 reference forwards), B=2.
    Reference: models/__init__.py:103-204."""
    from models import ddim_sampler
    torch.set_num_threads(8)
    net, d = ref_net(192, 192, 128, 1234)
    x_T, cond = rnd(831, 2, 15, 128, 128), rnd(832, 2, 6, 128, 128)
    out = ddim_sampler(x_T.clone(), net, cond=cond, subsample_steps=5, denoise=True, clip_before=True,
                       final_only=True, t_min=-1, log=True)
    o = out[0]
    save("traj_ddim_full", samples=o.reshape(2, -1)[:, ::30].clone(), first_row=o[:, :, 0, :].clone(),
         stats=torch.stack([o.mean(), o.std(), o.abs().max()]))


def gen_unet_ddpm():
    """The reference's alternative score network models/unet.py::UNet_DDPM (reduced: ngf 32, 32x32), with and without
    time conditioning, through its own forward and through the reference DDPM sampler."""
    from models.unet import UNet_DDPM
    from models import ddpm_sampler
    from oracle import unet_ddpm as OU
    out = {}
    for tc in (True, False):
        cfg = ref_config(32, 32, 32)
        cfg.model.time_conditional = tc
        net = UNet_DDPM(cfg).eval()
        d = OU.Dims(ngf=32, time_conditional=tc)
        p = OU.seeded_params(d, 61)
        own = {k: v for k, v in net.state_dict().items() if k.startswith("unet.")}
        assert set(own) == set(p), sorted(set(own) ^ set(p))[:8]
        for k, v in p.items():
            assert tuple(own[k].shape) == tuple(v.shape), k
        missing, unexpected = net.load_state_dict(p, strict=False)
        assert not unexpected and all(not m.startswith("unet.") for m in missing), missing
        x, cond = rnd(62, 2, 15, 32, 32), rnd(63, 2, 6, 32, 32)
        tag = "tc" if tc else "notc"
        with torch.no_grad():
            out[f"out_{tag}_t0"] = net(x, torch.tensor([0, 0]), cond=cond)
            out[f"out_{tag}_t500"] = net(x, torch.tensor([500, 500]), cond=cond)
        if tc:
            noises = [rnd(70 + i, 2, 15, 32, 32) for i in range(4)]
            it = iter(noises)
            orig = torch.randn_like
            torch.randn_like = lambda t, **kw: next(it)
            try:
                out["ddpm_tc"] = ddpm_sampler(x.clone(), net, cond=cond, subsample_steps=4, denoise=True, clip_before=True,
                                              final_only=True, t_min=-1, log=True)
            finally:
                torch.randn_like = orig
            out["alphas"] = net.alphas
    # the same network at ngf 64: its attention runs in one 128-wide head (a second kernel instantiation on the GPU side)
    cfg = ref_config(64, 64, 32)
    cfg.model.time_conditional = True
    net = UNet_DDPM(cfg).eval()
    p = OU.seeded_params(OU.Dims(ngf=64, time_conditional=True), 64)
    missing, unexpected = net.load_state_dict(p, strict=False)
    assert not unexpected and all(not m.startswith("unet.") for m in missing), missing
    with torch.no_grad():
        out["out_ngf64_t500"] = net(rnd(62, 1, 15, 32, 32), torch.tensor([500]), cond=rnd(63, 1, 6, 32, 32))
    # the wrapper options of UNet_DDPM (models/unet.py:337-372): noise_in_cond with the injected draw (mixed labels), and the
    # cosine schedule's buffers
    cfg = ref_config(32, 32, 32, noise_in_cond=True, sigma_dist="cosine")
    cfg.model.time_conditional = True
    net = UNet_DDPM(cfg).eval()
    net.load_state_dict(OU.seeded_params(OU.Dims(ngf=32, time_conditional=True), 61), strict=False)
    z = rnd(64, 2, 6, 32, 32)
    orig = torch.randn_like
    torch.randn_like = lambda t, **kw: z
    try:
        with torch.no_grad():
            out["out_nic_cos"] = net(x, torch.tensor([500, 7]), cond=cond)
    finally:
        torch.randn_like = orig
    out["cos_alphas"] = net.alphas
    save("unet_ddpm", **out)


def gen_forward_spade():
    """The SPADE-conditioned variant (``model.spade: true`` -> SPADE_NCSNpp, ncsnpp_more.py:396-718), reduced size:
    conditioning frames enter through per-act-norm gamma/beta maps instead of the input concat."""
    from models.better.ncsnpp_more import UNetMore_DDPM
    from oracle import scorenet_spade as OS
    cfg = ref_config(32, 32, 32)
    cfg.model.spade = True
    cfg.model.spade_dim = 32
    net = UNetMore_DDPM(cfg).eval()
    d = Dims(ngf=32, n_head_channels=32, image_size=32)
    p = OS.seeded_params(d, 81, spade_dim=32)
    own = dict(net.named_parameters())
    assert set(own) == set(p), sorted(set(own) ^ set(p))[:8]
    assert [k for k in own] == [k for k in p], "state_dict order"
    for k, v in p.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    missing, unexpected = net.load_state_dict(p, strict=False)
    assert not unexpected and all(m in ("betas", "alphas", "alphas_prev", "unet.sigmas", "k", "k_cum", "theta_t")
                                  for m in missing), missing
    x, cond = rnd(82, 2, 15, 32, 32), rnd(83, 2, 6, 32, 32)
    with torch.no_grad():
        o0 = net(x, torch.tensor([0, 0]), cond=cond)
        o1 = net(x, torch.tensor([990, 3]), cond=cond)
        o2 = net(x, torch.tensor([-0.5, -0.5]), cond=cond)
    save("forward_spade", out_t0=o0, out_t990_3=o1, out_tm05=o2)


def pseudo3d_dims():
    return Dims(ngf=32, ch_mult=[1, 2], num_res_blocks=1, attn_resolutions=[16, 8], n_head_channels=32, image_size=16,
                channels=3, num_frames=3, num_frames_cond=2)


def gen_forward_conv3d():
    """``model.arch: unetmore3d``: the same network with nn.Conv3d (3x3x3 / 1x1x1, layers3d.py:225-254) in place of the
    pseudo-3-D convolution pairs."""
    gen_forward_pseudo3d(arch="unetmore3d", name="forward_conv3d", seed=96)


def gen_forward_pseudo3d(arch="unetmorepseudo3d", name="forward_pseudo3d", seed=91):
    """``model.arch: unetmorepseudo3d`` (ncsnpp_more.py:40-51,101-122 + models/better/layers3d.py), reduced size: every 3x3 /
    1x1 convolution a per-frame Conv2d -> SiLU -> Conv1d over the frames, GroupNorm over (C / G, N, H, W), space-then-time
    attention, and the 1x1 "converters" from 5 frames (3 + 2 conditioning) to 3.  Outputs at three labels + strided samples
    of every module's output (forward hooks)."""
    from models.better.ncsnpp_more import UNetMore_DDPM
    from oracle import scorenet_pseudo3d as O3
    d = pseudo3d_dims()
    cfg = ref_config(d.ngf, d.n_head_channels, d.image_size, arch=arch, ch_mult=d.ch_mult,
                     num_res_blocks=d.num_res_blocks, attn_resolutions=d.attn_resolutions)
    cfg.data.num_frames, cfg.data.num_frames_cond, cfg.data.num_frames_future = d.num_frames, d.num_frames_cond, 0
    net = UNetMore_DDPM(cfg).eval()
    p = O3.seeded_params(d, seed, arch=arch)
    own = dict(net.named_parameters())
    assert set(own) == set(p), sorted(set(own) ^ set(p))[:8]
    assert [k for k in own] == [k for k in p], "state_dict order"
    for k, v in p.items():
        assert tuple(own[k].shape) == tuple(v.shape), (k, tuple(own[k].shape), tuple(v.shape))
    missing, unexpected = net.load_state_dict(p, strict=False)
    assert not unexpected and all(m in ("betas", "alphas", "alphas_prev", "unet.sigmas") for m in missing), missing
    B = 2
    x, cond = rnd(92, B, 3 * d.num_frames, 16, 16), rnd(93, B, 3 * d.num_frames_cond, 16, 16)
    taps = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, idx=idx: taps.__setitem__(idx, o.detach().clone()))
             for idx, m in enumerate(net.unet.all_modules)]
    with torch.no_grad():
        o1 = net(x, torch.tensor([430, 7]), cond=cond)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        o0 = net(x, torch.tensor([0, 0]), cond=cond)
        o2 = net(x, torch.tensor([-0.5, -0.5]), cond=cond)
    samples = {}
    for idx, t in taps.items():
        if idx < 2 or t.dim() != 4:
            continue
        flat = t.reshape(-1)
        stride = max(1, flat.numel() // 512)
        samples[f"tap{idx}"] = flat[::stride][:512].clone()
        samples[f"tapstat{idx}"] = torch.stack([t.mean(), t.std()])
        samples[f"tapshape{idx}"] = torch.tensor(t.shape)
    save(name, out_t0=o0, out_t430_7=o1, out_tm05=o2, **samples)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    gens = dict(schedule=gen_schedule, fir=gen_fir, blocks=gen_blocks, forward_reduced=gen_forward_reduced,
                samplers=gen_samplers, sampler_options=gen_sampler_options, sampler_gamma=gen_sampler_gamma, model_options=gen_model_options, lpips_lin=gen_lpips_lin,
                forward_full=gen_forward_full, forward_full_b9=gen_forward_full_b9,
                traj_full=gen_traj_full, traj_fpndm_full=gen_traj_fpndm_full, traj_ddim_full=gen_traj_ddim_full, unet_ddpm=gen_unet_ddpm, forward_spade=gen_forward_spade,
                forward_pseudo3d=gen_forward_pseudo3d, forward_conv3d=gen_forward_conv3d)
    for name, fn in gens.items():
        if a.only and name != a.only:
            continue
        if a.skip_full and name in ("forward_full", "forward_full_b9", "traj_full", "traj_fpndm_full", "traj_ddim_full"):
            continue
        fn()
