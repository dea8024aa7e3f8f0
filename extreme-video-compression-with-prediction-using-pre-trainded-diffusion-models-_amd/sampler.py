"""DDPM / DDIM / F-PNDM sampling loops on MI355X.

Same call surface as the reference samplers (models/__init__.py:39-100, 103-204, 207-342):
``sampler(x_mod, scorenet, cond=None, final_only=..., denoise=..., subsample_steps=..., clip_before=...,
same_noise=..., noise_val=..., **kw) -> Tensor (1, B, C, H, W)`` and they read
``scorenet.alphas / alphas_prev / betas`` (and ``.module`` if wrapped).  The per-step scalar coefficients
are formed on the host in float32 with exactly the reference's expressions; the tensor update of each
step is ONE fused elementwise kernel (evc_ddpm_step_f32 / evc_ddim_step_f32 / evc_pndm_transfer_f32)
instead of the reference's ~8 separate passes, and the 10 discarded logging norms
(models/__init__.py:297-303) are not computed.

Noise: ``noise_fn(i, x)`` injects the Gaussian the reference would draw with ``torch.randn_like`` after
step ``i`` (parity tests); otherwise it is drawn on-device from ``generator`` (Philox).
"""
import torch

from . import lib as L


def _net(scorenet):
    return scorenet.module if hasattr(scorenet, "module") else scorenet


def _eps(net, x, label, cond):
    if hasattr(net, "forward_label"):
        return net.forward_label(x, label, cond)
    labels = torch.full((x.shape[0],), label, device=x.device,
                        dtype=torch.long if float(label).is_integer() else torch.float32)
    return net(x, labels, cond=cond)


def _subsample(net, subsample_steps):
    """models/__init__.py:231-239 (float32 CPU arithmetic, incl. the lossy 1 - a/a_prev)."""
    alphas, alphas_prev, betas = net.alphas.cpu(), net.alphas_prev.cpu(), net.betas.cpu()
    steps = torch.arange(len(betas))
    if subsample_steps is not None and subsample_steps < len(alphas):
        skip = len(alphas) // subsample_steps
        steps = torch.tensor(list(range(0, len(alphas), skip)))
        alphas = alphas.index_select(0, steps)
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
        betas = 1.0 - torch.div(alphas, alphas_prev)
    return steps, alphas, alphas_prev, betas


def _gamma_tables(net, steps, full_len):
    """ks_cum / thetas of the ``gamma=True`` branch (models/__init__.py:226-227, 241-243): the model's Gamma-noise buffers,
    subsampled like the alphas.  A model built without ``config.model.gamma`` has none."""
    if not hasattr(net, "k_cum"):
        raise NotImplementedError("gamma=True needs the model's k_cum / theta_t buffers: build it with config.model.gamma = true "
                                  "(configs/mine.yml has gamma: false)")
    ks, th = net.k_cum.cpu(), net.theta_t.cpu()
    if len(steps) < full_len:
        ks, th = ks.index_select(0, steps), th.index_select(0, steps)
    return ks, th


def _gamma_noise(x, k, theta, alpha, raw=None, generator=None):
    """One Gamma-distributed noise tensor, standardised as the reference does (models/__init__.py:322-324):
    z ~ Gamma(k, rate 1/theta) per element, noise = (z - k theta) / sqrt(1 - alpha).  ``raw``: an injected z."""
    if raw is None:
        conc = torch.full(tuple(x.shape), float(k), device=x.device)
        # what torch.distributions.Gamma(conc, rate).sample() does (standard gamma / rate), with the caller's generator
        raw = torch._standard_gamma(conc, generator=generator) / float(1.0 / theta)
    raw = raw.to(x.device, torch.float32).contiguous()
    # z - k theta first, in fp32 like the reference: both are ~1e3 with a difference of order 1, so folding the subtraction
    # into one multiply-add with pre-divided constants would cost ~1e-4 of the noise
    centred = L.scale_clamp(raw, 1.0, -float(k * theta))
    return L.scale_clamp(centred, float(1.0 / (1 - alpha).sqrt()), 0.0, out=centred)


def _prepare(net, labels):
    if hasattr(net, "prepare_labels"):
        net.prepare_labels(labels)


def _run(gen):
    """Exhaust a step generator and return its value."""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


def run_interleaved(gens, streams):
    """Drive several independent sampler generators in lockstep, each on its own HIP stream: one network
    evaluation of every generator is enqueued per round, so the small latency-bound kernels and partial tile
    rounds of one group overlap the large convolutions of another.  Returns the generators' results in order.
    The caller's current stream must have been waited on by ``streams`` (inputs) and must wait on them after."""
    results = [None] * len(gens)
    live = list(range(len(gens)))
    while live:
        for i in list(live):
            with torch.cuda.stream(streams[i]):
                try:
                    next(gens[i])
                except StopIteration as e:
                    results[i] = e.value
                    live.remove(i)
    return results


@torch.no_grad()
def ddpm_sampler(*args, **kwargs):
    return _run(ddpm_steps(*args, **kwargs))


def ddpm_steps(x_mod, scorenet, cond=None, just_beta=False, final_only=False, denoise=True,
               subsample_steps=None, same_noise=False, noise_val=None, frac_steps=None, verbose=False,
               log=False, clip_before=True, t_min=-1, gamma=False, noise_fn=None, generator=None, **kwargs):
    """Generator form of ``ddpm_sampler``: yields after every network evaluation + update.

    ``frac_steps``: only the last fraction of the steps (models/__init__.py:248-256).  ``t_min`` > 0: ``x_mod`` is a clean
    previous frame; steps with label < t_min * (number of steps) are skipped and the first executed step first noises the
    input to its level, x <- sqrt(a_i) x + sqrt(1 - a_i) z (:266-277; z = ``noise_fn("t_min", x)`` when noise is injected).
    ``gamma``: the noise is a standardised Gamma sample of the model's k_cum / theta_t buffers (:226-227, :322-324; built only
    when config.model.gamma is set -- mine.yml has gamma: false -- otherwise NotImplementedError)."""
    net = _net(scorenet)
    steps, alphas, alphas_prev, betas = _subsample(net, subsample_steps)
    ks_cum = thetas = None
    if gamma:
        ks_cum, thetas = _gamma_tables(net, steps, len(net.betas))
    if frac_steps is not None:
        # the reference indexes the (possibly subsampled) tables with the step LABELS (:250-253): only consistent without
        # subsampling, where label == index; with it the reference raises IndexError, and so does this
        steps = steps[int((1 - frac_steps) * len(steps)):]
        idx = torch.as_tensor([int(v) for v in steps], dtype=torch.long)
        if len(idx) and int(idx.max()) >= len(alphas):
            raise IndexError("frac_steps with subsample_steps indexes the subsampled schedule by label (reference behaviour)")
        alphas, alphas_prev, betas = alphas[idx], alphas_prev[idx], betas[idx]
        if gamma:
            ks_cum, thetas = ks_cum[idx], thetas[idx]
    L_ = len(steps)
    run = [i for i, st in enumerate(steps) if not (int(st) < t_min * len(alphas))]
    _prepare(net, [int(steps[i]) for i in run] + ([L_ - 1] if denoise else []))
    x = x_mod.detach().to(torch.float32).clone().contiguous()
    if same_noise and noise_val is None:
        noise_val = x.clone()
    images = []
    x_transf = False

    def draw(tag, i):
        """The noise the reference draws at this point: Gaussian, or (gamma) a standardised Gamma sample of step i's
        parameters; an injected ``noise_fn`` supplies the RAW draw (randn / Gamma sample) in both cases."""
        raw = None if noise_fn is None else noise_fn(tag, x)
        if gamma:
            return _gamma_noise(x, ks_cum[i], thetas[i], alphas[i], raw=raw, generator=generator)
        if raw is not None:
            return raw.to(x.device, torch.float32).contiguous()
        return torch.randn(x.shape, device=x.device, dtype=torch.float32, generator=generator)
    for i in run:
        step = steps[i]
        if not x_transf and t_min > 0:          # noise the clean input to this step's level
            x = L.lincomb4([x, draw("t_min", i)], [float(alphas[i].sqrt()), float((1 - alphas[i]).sqrt())])
        x_transf = True
        c_beta, c_alpha, c_alpha_prev = betas[i], alphas[i], alphas_prev[i]
        e = _eps(net, x, int(step), cond)
        k1 = float(1 / c_alpha.sqrt())
        k2 = float((1 - c_alpha).sqrt())
        c1 = float(c_alpha_prev.sqrt() * c_beta / (1 - c_alpha))
        c2 = float((1 - c_beta).sqrt() * (1 - c_alpha_prev) / (1 - c_alpha))
        noise, sigma = None, 0.0
        if i + 1 != L_:
            noise = noise_val if same_noise else draw(i, i)
            sigma = float(c_beta.sqrt()) if just_beta else float(((1 - c_alpha_prev) / (1 - c_alpha) * c_beta).sqrt())
        L.ddpm_step(x, e, noise, k1, k2, c1, c2, sigma, clip_before)
        if not final_only:
            images.append(x.to("cpu"))
        yield
    if denoise:
        e = _eps(net, x, L_ - 1, cond)   # label is the COUNT index L-1 (models/__init__.py:333-335)
        x = L.axpy(x, e, -float((1 - alphas[-1]).sqrt()))
        if not final_only:
            images.append(x.to("cpu"))
        yield
    return x.unsqueeze(0) if final_only else torch.stack(images)


@torch.no_grad()
def ddim_sampler(*args, **kwargs):
    return _run(ddim_steps(*args, **kwargs))


def ddim_steps(x_mod, scorenet, cond=None, final_only=False, denoise=True, subsample_steps=None, verbose=False,
               log=True, clip_before=True, t_min=-1, gamma=False, **kwargs):
    """Generator form of ``ddim_sampler``; ``t_min`` / ``gamma`` as in ``ddpm_steps`` (models/__init__.py:145-157; DDIM draws
    noise only for the t_min start)."""
    noise_fn, generator = kwargs.get("noise_fn"), kwargs.get("generator")
    net = _net(scorenet)
    steps, alphas, alphas_prev, betas = _subsample(net, subsample_steps)
    ks_cum = thetas = None
    if gamma:
        ks_cum, thetas = _gamma_tables(net, steps, len(net.betas))
    L_ = len(steps)
    run = [i for i, st in enumerate(steps) if not (int(st) < t_min * len(alphas))]
    _prepare(net, [int(steps[i]) for i in run] + ([L_ - 1] if denoise else []))
    x = x_mod.detach().to(torch.float32).clone().contiguous()
    images = []
    x_transf = False
    for i in run:
        step = steps[i]
        if not x_transf and t_min > 0:
            raw = None if noise_fn is None else noise_fn("t_min", x)
            if gamma:
                z = _gamma_noise(x, ks_cum[i], thetas[i], alphas[i], raw=raw, generator=generator)
            elif raw is not None:
                z = raw.to(x.device, torch.float32).contiguous()
            else:
                z = torch.randn(x.shape, device=x.device, dtype=torch.float32, generator=generator)
            x = L.lincomb4([x, z], [float(alphas[i].sqrt()), float((1 - alphas[i]).sqrt())])
        x_transf = True
        c_alpha, c_alpha_prev = alphas[i], alphas_prev[i]
        e = _eps(net, x, int(step), cond)
        L.ddim_step(x, e, float(1 / c_alpha.sqrt()), float((1 - c_alpha).sqrt()), float(c_alpha_prev.sqrt()),
                    float((1 - c_alpha_prev).sqrt()), clip_before)
        if not final_only:
            images.append(x.to("cpu"))
        yield
    if denoise:
        e = _eps(net, x, L_ - 1, cond)
        x = L.axpy(x, e, -float((1 - alphas[-1]).sqrt()))
        if not final_only:
            images.append(x.to("cpu"))
        yield
    return x.unsqueeze(0) if final_only else torch.stack(images)


def _transfer(x, t, t_next, e, alphas_cump, clip_before):
    """models/pndm.py:19-33 with the scalar coefficients formed on the host."""
    at = alphas_cump[int(t) + 1]          # .long() truncates toward zero, as int() does
    an = alphas_cump[int(t_next) + 1]
    d = float(an - at)
    cx = float(1 / (at.sqrt() * (at.sqrt() + an.sqrt())))
    ce = float(1 / (at.sqrt() * (((1 - an) * at).sqrt() + ((1 - at) * an).sqrt())))
    return L.pndm_transfer(x, e, d, cx, ce, clip_before)


@torch.no_grad()
def FPNDM_sampler(*args, **kwargs):
    return _run(fpndm_steps(*args, **kwargs))


def fpndm_steps(x_mod, scorenet, cond=None, final_only=False, denoise=True, subsample_steps=None, verbose=False,
                log=True, clip_before=True, t_min=-1, gamma=False, **kwargs):
    """Generator form of ``FPNDM_sampler`` (yields once per outer iteration).  FPNDM_sampler + pndm.gen_order_4 / runge_kutta (models/__init__.py:39-100, models/pndm.py:3-52):
    3 Runge-Kutta warm-up iterations (4 evaluations each) then 4-term Adams-Bashforth; no denoise call."""
    net = _net(scorenet)
    alphas_old = net.alphas.cpu().flip(0)
    skip = len(alphas_old) // subsample_steps
    steps = list(range(0, len(alphas_old), skip))
    steps_next = [-1] + steps[:-1]
    _prepare(net, fpndm_labels(len(alphas_old), subsample_steps))
    x = x_mod.detach().to(torch.float32).clone().contiguous()
    ets, images = [], []
    for t, tn in zip(steps, steps_next):
        if len(ets) > 2:
            ets.append(_eps(net, x, t, cond))
            e = L.lincomb4([ets[-1], ets[-2], ets[-3], ets[-4]], [55 / 24, -59 / 24, 37 / 24, -9 / 24])
            ets = ets[-4:]   # the reference keeps the whole history (never trimmed); only 4 are ever read
        else:
            tm = (t + tn) / 2
            e1 = _eps(net, x, t, cond)
            ets.append(e1)
            x2 = _transfer(x, t, tm, e1, alphas_old, clip_before)
            e2 = _eps(net, x2, tm, cond)
            x3 = _transfer(x, t, tm, e2, alphas_old, clip_before)
            e3 = _eps(net, x3, tm, cond)
            x4 = _transfer(x, t, tn, e3, alphas_old, clip_before)
            e4 = _eps(net, x4, tn, cond)
            e = L.lincomb4([e1, e2, e3, e4], [1 / 6, 2 / 6, 2 / 6, 1 / 6])
        x = _transfer(x, t, tn, e, alphas_old, clip_before)
        if not final_only:
            images.append(x.to("cpu"))
        yield
    return x.unsqueeze(0) if final_only else torch.stack(images)


def fpndm_labels(n_classes, subsample_steps):
    """Every label F-PNDM passes to the network: the step values, the Runge-Kutta midpoints of the first three
    iterations and -1 (models/__init__.py:59-92, models/pndm.py:3-17)."""
    skip = n_classes // subsample_steps
    steps = list(range(0, n_classes, skip))
    steps_next = [-1] + steps[:-1]
    labels = set()
    for i, (t, tn) in enumerate(zip(steps, steps_next)):
        labels.add(float(t))
        if i < 3:
            labels.update([(t + tn) / 2, float(tn)])
    return sorted(labels)


def label_set(sampler_fn, scorenet, subsample_steps=None, denoise=True):
    """The exact set of network labels ``sampler_fn`` will use.  ``ClipDecoder.generate`` evaluates their AdaGN
    table rows on the main stream BEFORE clip groups fan out to their own streams (the table is shared state; a
    row built lazily on one group's stream would be read by the others with no ordering)."""
    net = _net(scorenet)
    n = len(net.betas)
    if sampler_fn is FPNDM_sampler:
        return fpndm_labels(n, subsample_steps or n)
    steps, _, _, _ = _subsample(net, subsample_steps)
    return [float(v) for v in steps] + ([float(len(steps) - 1)] if denoise else [])



def get_step_generator(sampler_fn):
    """The generator form of a sampler returned by ``get_sampler`` (None for foreign callables)."""
    return {ddpm_sampler: ddpm_steps, ddim_sampler: ddim_steps, FPNDM_sampler: fpndm_steps}.get(sampler_fn)


def get_sampler(version):
    """city_sender.py:248-254 (+ FPNDM for BASELINE config 5)."""
    version = version.upper()
    if version == "DDPM":
        return ddpm_sampler
    if version == "DDIM":
        return ddim_sampler
    if version in ("FPNDM", "PNDM"):
        return FPNDM_sampler
    raise ValueError(f"unknown sampler {version}")
