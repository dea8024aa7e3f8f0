"""Oracle: noise schedule and time-step subsampling (test infrastructure only).

Follows reference ``models/__init__.py:17-36`` (get_sigmas), ``models/better/ncsnpp_more.py:735-739``
(betas / alphas / alphas_prev buffers) and the subsampling blocks of the samplers
(``models/__init__.py:122-135`` and ``:231-239``).
"""
import numpy as np
import torch


def base_schedule(sigma_begin=0.02, sigma_end=1e-4, num_classes=1000):
    """betas[0] is the NOISIEST level; alphas is the reversed cumulative product."""
    betas = torch.linspace(sigma_begin, sigma_end, num_classes)          # models/__init__.py:25-27
    alphas = torch.cumprod(1 - betas.flip(0), 0).flip(0)                  # ncsnpp_more.py:737
    alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])  # ncsnpp_more.py:738
    return betas, alphas, alphas_prev


def cosine_schedule(num_classes=1000):
    """sigma_dist 'cosine': get_sigmas (models/__init__.py:29-33) are the alphas, betas derived (ncsnpp_more.py:740-743)."""
    import numpy as np
    T = num_classes
    t = torch.linspace(T, 0, T + 1) / T
    f = torch.cos((t + 0.008) / (1 + 0.008) * np.pi / 2) ** 2
    alphas = f[:-1] / f[-1]
    alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
    return 1 - alphas / alphas_prev, alphas, alphas_prev


def subsample(alphas, alphas_prev, betas, subsample_steps):
    """models/__init__.py:231-239: returns (steps, alphas, alphas_prev, betas) after subsampling."""
    steps = torch.arange(len(betas))
    if subsample_steps is not None and subsample_steps < len(alphas):
        skip = len(alphas) // subsample_steps
        steps = torch.tensor(list(range(0, len(alphas), skip)))
        alphas = alphas.index_select(0, steps)
        alphas_prev = torch.cat([alphas[1:], torch.tensor([1.0]).to(alphas)])
        betas = 1.0 - torch.div(alphas, alphas_prev)
    return steps, alphas, alphas_prev, betas


def gamma_schedule(betas, alphas, theta_0=0.001):
    """ncsnpp_more.py:744-749: the (k, k_cum, theta_t) buffers a ``config.model.gamma`` model registers."""
    k = betas / (alphas * (theta_0 ** 2))
    k_cum = torch.cumsum(k.flip(0), 0).flip(0)
    theta_t = torch.sqrt(alphas) * theta_0
    return k, k_cum, theta_t
