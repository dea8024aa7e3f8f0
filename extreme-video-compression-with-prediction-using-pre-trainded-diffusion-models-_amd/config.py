"""configs/mine.yml loader with the reference's ``--config_mod`` override grammar
(city_sender.py:138-170, function.py:24-32).  The YAML schema is loaded unchanged; overrides are parsed
with ``ast.literal_eval`` instead of the reference's ``eval``."""
import argparse
import ast

import yaml

DEFAULT_CONFIG_MOD = "model.ngf=192 model.n_head_channels=192"


def dict2namespace(config):
    ns = argparse.Namespace()
    for k, v in config.items():
        setattr(ns, k, dict2namespace(v) if isinstance(v, dict) else v)
    return ns


def namespace2dict(ns):
    return {k: namespace2dict(v) if isinstance(v, argparse.Namespace) else v for k, v in vars(ns).items()}


def apply_config_mod(config, config_mod):
    """``"section.key=value section.key=value"`` (also accepts a list of such strings)."""
    if not config_mod:
        return config
    if isinstance(config_mod, (list, tuple)):
        config_mod = " ".join(config_mod)
    for val in config_mod.split(" "):
        if not val:
            continue
        key, config_val = val.split("=")
        section, name = key.split(".")
        cur = config[section][name]
        try:
            totest = cur[0]
        except Exception:
            totest = cur
        if isinstance(totest, str):
            config[section][name] = config_val
        else:
            config[section][name] = ast.literal_eval(config_val)
    return config


def load_config(path, config_mod=DEFAULT_CONFIG_MOD):
    with open(path, "r") as f:
        config = yaml.safe_load(f)
    config = apply_config_mod(config, config_mod)
    if config["model"].get("output_all_frames", False):
        config["model"]["noise_in_cond"] = True
    return dict2namespace(config), config


def default_config(ngf=192, n_head_channels=192, image_size=128, subsample=100):
    """The live keys of configs/mine.yml as a namespace, for callers without the file."""
    cfg = {
        "sampling": {"subsample": subsample, "denoise": True, "clip_before": True, "init_prev_t": -1.0,
                     "n_steps_each": 0, "step_lr": 0.0, "num_frames_pred": 28, "preds_per_test": 1, "ckpt_id": 0},
        "data": {"dataset": "Cityscapes", "image_size": image_size, "channels": 3, "num_frames": 5,
                 "num_frames_cond": 2, "num_frames_future": 0, "rescaled": True, "logit_transform": False,
                 "uniform_dequantization": False, "gaussian_dequantization": False},
        "model": {"version": "DDPM", "gamma": False, "arch": "unetmore", "type": "v1", "time_conditional": True,
                  "dropout": 0.0, "sigma_dist": "linear", "sigma_begin": 0.02, "sigma_end": 0.0001,
                  "num_classes": 1000, "ema": True, "ema_rate": 0.999, "ngf": ngf, "ch_mult": [1, 1, 2, 3, 4],
                  "num_res_blocks": 2, "attn_resolutions": [8, 16, 32], "n_head_channels": n_head_channels,
                  "noise_in_cond": False, "output_all_frames": False, "cond_emb": False, "spade": False},
    }
    return dict2namespace(cfg)
