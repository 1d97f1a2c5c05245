"""Hydra-compatible configuration loading for the reference's ``configs/`` tree without Hydra.

The reference composes ``configs/main_config.yaml`` through ``@hydra.main`` (main_final.py:751): a ``defaults`` list
(``data: data_final``, ``model: unet_convlstm_attention``, ``training: default``, ``trainer: default``, ``_self_``),
group files headed ``# @package _global_.<group>``, and dotted CLI overrides (commands.md:3-11).  hydra/omegaconf
are not installed in this image, so this module implements exactly that subset on PyYAML and returns an
attribute-access dict; when Hydra is importable the reference's own entry point can be used instead.
"""
import copy
import os
import re

import yaml


class Cfg(dict):
    """dict with attribute access (enough of DictConfig for get_model / the trainer)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return super().get(k, default)


def _wrap(o):
    if isinstance(o, dict):
        return Cfg({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


_FLOAT = re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$")


def _parse_scalar(s: str):
    v = yaml.safe_load(s)
    if isinstance(v, str) and _FLOAT.match(v):     # PyYAML reads "5e-4" as a string; Hydra/OmegaConf as a float
        return float(v)
    return v


def _fix_floats(o):
    if isinstance(o, dict):
        return {k: _fix_floats(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_fix_floats(v) for v in o]
    if isinstance(o, str) and _FLOAT.match(o):
        return float(o)
    return o


def load_config(config_dir: str, config_name: str = "main_config.yaml", overrides=()):
    """Compose ``config_dir/config_name`` with its defaults list, then apply ``key.sub=value`` / ``group=name``."""
    with open(os.path.join(config_dir, config_name)) as f:
        main = yaml.safe_load(f) or {}
    defaults = main.pop("defaults", [])
    main.pop("hydra", None)
    group_choice = {}
    for d in defaults:
        if isinstance(d, dict):
            group_choice.update(d)
    scalar_over = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        k = k.lstrip("+")
        if k in group_choice and "." not in k:
            group_choice[k] = v
        else:
            scalar_over.append((k, v))
    cfg = {}
    for group, name in group_choice.items():
        path = os.path.join(config_dir, group, f"{name}.yaml")
        if not os.path.exists(path):
            raise FileNotFoundError(f"config group file not found: {path}")
        with open(path) as f:
            cfg[group] = _fix_floats(yaml.safe_load(f) or {})
    for k, v in _fix_floats(copy.deepcopy(main)).items():       # _self_ last
        cfg[k] = v
    for k, v in scalar_over:
        node = cfg
        parts = k.split(".")
        for part in parts[:-1]:
            node = node.setdefault(part, {})
        node[parts[-1]] = _parse_scalar(v)
    return _wrap(cfg)


def synthetic_config(base_channels=32, seq_len=6, n_inputs=5, n_outputs=2, lr=5e-4, weight_decay=0.0):
    """The BASELINE.json benchmark configuration as a config object (same keys as the reference's YAML tree)."""
    return _wrap({
        "data": {"input_vars": ["CO2", "SO2", "CH4", "BC", "rsdt"][:n_inputs], "output_vars": ["tas", "pr"][:n_outputs],
                 "seq_len": seq_len, "batch_size": 32},
        "model": {"type": "unet_convlstm_attention", "base_channels": base_channels},
        "training": {"lr": lr, "weight_decay": weight_decay},
        "trainer": {"max_epochs": 50, "accelerator": "cuda", "devices": 1, "precision": 32},
        "seed": 42,
    })
