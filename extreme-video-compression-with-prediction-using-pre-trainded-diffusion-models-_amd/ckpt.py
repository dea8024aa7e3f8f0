"""Checkpoint layouts of the reference, read with loaders that execute nothing from the file.

Diffusion: ``checkpoints/sender/checkpoint_<id>.pt`` is a Python LIST ``states``:
``states[0]`` = model ``state_dict`` saved from an ``nn.DataParallel`` wrapper (``module.`` prefix),
``states[-1]`` = EMA shadow ``{param_name: tensor}`` without prefix, copied over the parameters when
``model.ema`` is true (city_sender.py:304-324, models/ema.py:23-28,46-57).
ELIC: ``checkpoints/neural network/<q>.pth.tar`` is a flat state dict; compressai's
``zoo.load_state_dict`` strips ``module.`` and renames legacy entropy-bottleneck keys.
"""
import re

import torch


def load_diffusion_checkpoint(path, ema=True, map_location="cpu"):
    states = torch.load(path, map_location=map_location, weights_only=True)
    return resolve_diffusion_states(states, ema)


def resolve_diffusion_states(states, ema=True):
    if not isinstance(states, (list, tuple)) or len(states) < 1:
        raise ValueError("diffusion checkpoint must be the reference's list [model_state, ..., ema_shadow]")
    sd = {}
    for k, v in states[0].items():
        sd[k[len("module."):] if k.startswith("module.") else k] = v
    if ema:
        shadow = states[-1]
        if not isinstance(shadow, dict):
            raise ValueError("states[-1] is not an EMA shadow dict")
        for k, v in shadow.items():   # EMAHelper.ema(): param.data.copy_(shadow[name])
            if k in sd:
                sd[k] = v
    return sd


def make_diffusion_states(state_dict, ema_state_dict=None):
    """Inverse of ``resolve_diffusion_states`` (used to synthesise reference-layout checkpoints)."""
    model = {"module." + k: v for k, v in state_dict.items()}
    shadow = dict(ema_state_dict if ema_state_dict is not None else
                  {k: v for k, v in state_dict.items() if k.startswith("unet.")})
    return [model, {"step": 0}, shadow]


_LEGACY = re.compile(r"^(entropy_bottleneck)\._(biases|matrices|factors)\.(\d+)$")


def load_elic_state_dict(path, map_location="cpu"):
    sd = torch.load(path, map_location=map_location, weights_only=True)
    if "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return normalise_elic_state_dict(sd)


def normalise_elic_state_dict(sd):
    out = {}
    names = {"biases": "_bias", "matrices": "_matrix", "factors": "_factor"}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        m = _LEGACY.match(k)
        if m:
            k = f"{m.group(1)}.{names[m.group(2)]}{m.group(3)}"
        out[k] = v
    return out
