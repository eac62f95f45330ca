"""The reference's run configuration -> this package's objects: what `run_pretrain.py` does between reading its JSON and the first
training step, for the hot path only.

Reference: `src/configs/config.py:12-30` (`parse_with_config`: the JSON's keys become attributes, nested dicts too),
`src/pretrain/run_pretrain.py:109-126` (`setup_model`), `:234-236` (`amp.initialize(model, optimizer, enabled=cfg.fp16,
opt_level=cfg.amp_level)`), `:324` (`build_loss_func(cfg.loss_config)`), `:222` (`setup_e2e_optimizer(model, cfg)`), `:408-423` (schedule
+ clipping), and the shipped `src/configs/pretrain/pretrain_vip_base_16.json`.

What changes, and only this: `fp16: 1` with apex `amp_level` O1 / O2 ("half-precision compute, fp32 master weights, dynamic loss scale")
becomes bf16 compute on fp32 masters WITHOUT loss scaling (`set_compute_dtype(torch.bfloat16)`; `fp16: 0` -> float32); hub names
(`openai/clip-vit-base-patch16`) resolve to a local directory of that name if there is one, else -- there is no hub access -- to the
architecture's dimensions (`KNOWN_CLIP_CONFIGS`), and weights that are not on disk are an error unless random init is asked for.
"""
from __future__ import annotations

import json
import os

import torch

from . import workload
from .modeling import VidCLIP
from .optimization import build_loss_func, get_lr_sched, setup_e2e_optimizer
from .utils.load_save import load_state_dict_with_mismatch

# config.json dimensions of the hub names the shipped configs use (the files themselves are not redistributed here)
KNOWN_CLIP_CONFIGS = {"openai/clip-vit-base-patch16": lambda: workload.vit_b_config(16),
                      "openai/clip-vit-base-patch32": lambda: workload.vit_b_config(32)}


class Cfg(dict):
    """attribute access on a (nested) dict -- what the reference gets from `easydict.EasyDict` (config.py:21)"""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, _wrap(v))

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, Cfg):
        return Cfg(v)
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def load_config(src, **overrides) -> Cfg:
    """`parse_with_config` without the argparse half: a JSON file (or a dict) -> attribute-access config; keyword overrides play the
    command line's role (config.py:23-28: the command line wins)."""
    if isinstance(src, (str, os.PathLike)):
        with open(src) as fh:
            src = json.load(fh)
    return Cfg(src, **overrides)


def compute_dtype(cfg) -> torch.dtype:
    """run_pretrain.py:234-236 -> the compute dtype of the HIP path (module docstring)"""
    if not int(getattr(cfg, "fp16", 0)):
        return torch.float32
    level = str(getattr(cfg, "amp_level", "O2"))
    if level not in ("O1", "O2"):
        raise ValueError(f"amp_level={level!r}: the reference runs O1 / O2 (O0 is fp16: 0, O3 -- no fp32 masters -- is not supported)")
    return torch.bfloat16


def _resolve_clip_config(name):
    if isinstance(name, dict) or os.path.isdir(str(name)) or os.path.isfile(str(name)):
        return name
    if name in KNOWN_CLIP_CONFIGS:
        return KNOWN_CLIP_CONFIGS[name]()
    raise FileNotFoundError(f"clip_config={name!r}: not a local directory / config.json and not one of {sorted(KNOWN_CLIP_CONFIGS)} "
                            "(there is no hub access)")


def setup_model(cfg, device=None, allow_random_init=False) -> VidCLIP:
    """run_pretrain.py:109-126 + the precision half of :234-236"""
    args = Cfg(cfg)
    args.clip_config = _resolve_clip_config(cfg.clip_config)
    weights = getattr(cfg, "clip_weights", None)
    if weights and not os.path.exists(str(weights)):
        if not allow_random_init:
            raise FileNotFoundError(f"clip_weights={weights!r} is not on disk (no hub access); pass a local directory / state dict, or "
                                    "allow_random_init=True")
        weights = ""
    args.clip_weights = weights or ""
    model = VidCLIP(args)
    if getattr(cfg, "e2e_weights_path", None):
        load_state_dict_with_mismatch(model, cfg.e2e_weights_path)
    if getattr(cfg, "freeze_text_model", 0):
        model.freeze_text_encoder(bool(getattr(cfg, "freeze_text_proj", 0)))
    model.clipmodel.set_compute_dtype(compute_dtype(cfg))
    if device is not None:
        model.to(device)
    return model


def setup_training(cfg, model, num_train_steps: int):
    """loss, optimizer and schedule of run_pretrain.py:222, 324, 408-423 from the config: returns
    `Cfg(loss_fn, optimizer, lr_at(global_step), grad_norm)`; the loop sets `group['lr'] = lr_at(step)` on every parameter group
    (x lr_mul for the groups of `lr_mul_prefix`, as the reference does) and calls `optimizer.clip_and_step(grad_norm)`."""
    optimizer = setup_e2e_optimizer(model, cfg)
    decay, lr, warm = getattr(cfg, "decay", "linear"), cfg.learning_rate, getattr(cfg, "warmup_ratio", 0.1)

    def lr_at(global_step):
        return get_lr_sched(global_step, decay, lr, num_train_steps, warmup_ratio=warm)
    return Cfg(loss_fn=build_loss_func(cfg.loss_config), optimizer=optimizer, lr_at=lr_at, grad_norm=float(getattr(cfg, "grad_norm", -1)))
