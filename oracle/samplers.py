"""Oracle: DDPM / DDIM / F-PNDM sampling loops (test infrastructure only).

Restates reference ``models/__init__.py:207-342`` (ddpm_sampler), ``:103-204`` (ddim_sampler),
``:39-100`` (FPNDM_sampler) and ``models/pndm.py:3-52`` for ``gamma=False``, ``just_beta=False``, ``final_only=True``
(the CLI's settings), plus the ``t_min`` / ``frac_steps`` / ``gamma`` options of the DDPM / DDIM loops.

``eps_fn(x, labels)`` is the score network with ``cond`` already bound; ``labels`` is an int64
(DDPM/DDIM) or float (F-PNDM) tensor of shape (B,).  ``noise_fn(i, x)`` supplies the Gaussian noise
added after step ``i`` (the reference draws ``torch.randn_like``; parity tests inject it).  With ``gamma=(k_cum, theta_t)``
(``schedule.gamma_schedule``) the draw is a Gamma(k_cum_i, rate 1/theta_i) sample instead -- ``noise_fn`` then supplies that
RAW sample -- standardised as ``(z - k theta) / sqrt(1 - alpha)`` (models/__init__.py:150-153, :275-278, :321-324).
"""
import torch

from .schedule import base_schedule, subsample


def _draw(tag, i, x, noise_fn, gamma_tabs, alphas):
    if gamma_tabs is None:
        return noise_fn(tag, x) if noise_fn is not None else torch.randn_like(x)
    ks_cum, thetas = gamma_tabs
    if noise_fn is not None:
        z = noise_fn(tag, x)
    else:
        z = torch.distributions.gamma.Gamma(torch.full(x.shape[1:], ks_cum[i]),
                                            torch.full(x.shape[1:], 1 / thetas[i])).sample((x.shape[0],)).to(x.device)
    return (z - ks_cum[i] * thetas[i]) / ((1 - alphas[i]).sqrt())


def _gamma_tabs(gamma, steps, full_len):
    if gamma is None:
        return None
    ks_cum, thetas = gamma
    if len(steps) < full_len:                                                         # :241-243
        idx = torch.as_tensor([int(v) for v in steps], dtype=torch.long)
        ks_cum, thetas = ks_cum.index_select(0, idx), thetas.index_select(0, idx)
    return ks_cum, thetas


def _labels(value, x, long=True):
    t = value * torch.ones(x.shape[0], device=x.device)
    return t.long() if long else t


@torch.no_grad()
def ddpm(x, eps_fn, sched, subsample_steps=None, denoise=True, clip_before=True, noise_fn=None, t_min=-1, frac_steps=None,
         gamma=None):
    betas, alphas, alphas_prev = sched
    full_len = len(betas)
    steps, alphas, alphas_prev, betas = subsample(alphas, alphas_prev, betas, subsample_steps)
    gamma = _gamma_tabs(gamma, steps, full_len)
    if frac_steps is not None:                                                        # :248-258 (tables indexed by LABEL)
        steps = steps[int((1 - frac_steps) * len(steps)):]
        idx = torch.as_tensor([int(v) for v in steps], dtype=torch.long)
        alphas, alphas_prev, betas = alphas[idx], alphas_prev[idx], betas[idx]
        if gamma is not None:
            gamma = (gamma[0][idx], gamma[1][idx])
    L = len(steps)
    x_transf = False
    for i, step in enumerate(steps):
        if step < t_min * len(alphas):                                                # :263-264
            continue
        if not x_transf and t_min > 0:                                                # :266-276
            z = _draw("t_min", i, x, noise_fn, gamma, alphas)
            x = alphas[i].sqrt() * x + (1 - alphas[i]).sqrt() * z
        x_transf = True
        c_beta, c_alpha, c_alpha_prev = betas[i], alphas[i], alphas_prev[i]
        grad = eps_fn(x, _labels(step, x))                                            # :285-286
        x0 = (1 / c_alpha.sqrt()) * (x - (1 - c_alpha).sqrt() * grad)                 # :289
        if clip_before:
            x0 = x0.clip_(-1, 1)
        x = (c_alpha_prev.sqrt() * c_beta / (1 - c_alpha)) * x0 + \
            ((1 - c_beta).sqrt() * (1 - c_alpha_prev) / (1 - c_alpha)) * x            # :292
        if i + 1 == L:                                                                # :313-315
            continue
        noise = _draw(i, i, x, noise_fn, gamma, alphas)                               # :321-326
        x = x + ((1 - c_alpha_prev) / (1 - c_alpha) * c_beta).sqrt() * noise          # :330
    if denoise:                                                                       # :333-335
        x = x - (1 - alphas[-1]).sqrt() * eps_fn(x, _labels(L - 1, x))
    return x.unsqueeze(0)


@torch.no_grad()
def ddim(x, eps_fn, sched, subsample_steps=None, denoise=True, clip_before=True, t_min=-1, noise_fn=None, gamma=None):
    betas, alphas, alphas_prev = sched
    full_len = len(betas)
    steps, alphas, alphas_prev, betas = subsample(alphas, alphas_prev, betas, subsample_steps)
    gamma = _gamma_tabs(gamma, steps, full_len)
    L = len(steps)
    x_transf = False
    for i, step in enumerate(steps):
        if step < t_min * len(alphas):                                                # :145-146
            continue
        if not x_transf and t_min > 0:                                                # :148-156
            z = _draw("t_min", i, x, noise_fn, gamma, alphas)
            x = alphas[i].sqrt() * x + (1 - alphas[i]).sqrt() * z
        x_transf = True
        c_alpha, c_alpha_prev = alphas[i], alphas_prev[i]
        grad = eps_fn(x, _labels(step, x))
        x0 = (1 / c_alpha.sqrt()) * (x - (1 - c_alpha).sqrt() * grad)                 # :163
        if clip_before:
            x0 = x0.clip_(-1, 1)
        x = c_alpha_prev.sqrt() * x0 + (1 - c_alpha_prev).sqrt() * grad               # :166
    if denoise:                                                                       # :195-197
        x = x - (1 - alphas[-1]).sqrt() * eps_fn(x, _labels(L - 1, x))
    return x.unsqueeze(0)


def _transfer(x, t, t_next, et, alphas_cump, clip_before):
    """pndm.py:19-33.  ``.long()`` truncates toward zero, so -0.5 -> 0."""
    at = alphas_cump[t.long() + 1].view(-1, 1, 1, 1)
    at_next = alphas_cump[t_next.long() + 1].view(-1, 1, 1, 1)
    x_delta = (at_next - at) * ((1 / (at.sqrt() * (at.sqrt() + at_next.sqrt()))) * x -
                                1 / (at.sqrt() * (((1 - at_next) * at).sqrt() + ((1 - at) * at_next).sqrt())) * et)
    x_next = x + x_delta
    if clip_before:
        x_next = x_next.clip_(-1, 1)
    return x_next


@torch.no_grad()
def fpndm(x, eps_fn, sched, subsample_steps, clip_before=True, labels_log=None):
    """FPNDM_sampler + pndm.gen_order_4; no denoise call, no noise."""
    betas, alphas, alphas_prev = sched
    alphas_old = alphas.flip(0)                                                       # :59
    skip = len(alphas) // subsample_steps
    steps = list(range(0, len(alphas), skip))
    steps_next = [-1] + steps[:-1]                                                    # :64

    def model(xx, t):
        if labels_log is not None:
            labels_log.append(float(t[0]))
        return eps_fn(xx, t)

    ets = []
    for i in range(len(steps)):
        t = _labels(steps[i], x)
        t_next = _labels(steps_next[i], x)
        if len(ets) > 2:                                                              # pndm.py:44-47
            ets.append(model(x, t))
            noise = (1 / 24) * (55 * ets[-1] - 59 * ets[-2] + 37 * ets[-3] - 9 * ets[-4])
        else:                                                                         # runge_kutta, pndm.py:3-17
            t_mid = (t + t_next) / 2
            e_1 = model(x, t)
            ets.append(e_1)
            x_2 = _transfer(x, t, t_mid, e_1, alphas_old, clip_before)
            e_2 = model(x_2, t_mid)
            x_3 = _transfer(x, t, t_mid, e_2, alphas_old, clip_before)
            e_3 = model(x_3, t_mid)
            x_4 = _transfer(x, t, t_next, e_3, alphas_old, clip_before)
            e_4 = model(x_4, t_next)
            noise = (1 / 6) * (e_1 + 2 * e_2 + 2 * e_3 + e_4)
        x = _transfer(x, t, t_next, noise, alphas_old, clip_before)
    return x.unsqueeze(0)


__all__ = ["ddpm", "ddim", "fpndm", "base_schedule"]
