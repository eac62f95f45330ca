"""Checkpoint I/O with the reference's on-disk conventions (``src/utils/load_save.py``), host-side only.

* ``ModelSaver`` (:38-63): ``{prefix}_step_{N}.pt`` is the model's plain ``state_dict()`` on CPU at full precision --
  for ``VidCLIP`` the 402 ``clipmodel.*`` tensors -- and, with an optimizer, ``{prefix}_step_{N}_train_state.pt`` =
  ``{'step', 'optimizer'}``.
* ``load_state_dict_with_mismatch`` (:86-115): in-place, non-strict load that skips tensors whose shape differs (a
  checkpoint trained with another ``temporal_size`` / resolution) and reports the three key sets.
* ``E2E_TrainingRestorer`` (:260-327): ``restore.pt`` + ``restore_backup.pt`` holding ``global_step``,
  ``model_state_dict`` and ``optim_state_dict`` with floating tensors stored as fp16 on CPU (:177-192) and widened back
  to fp32 on load (:159-174).  There is no ``amp_state_dict`` here (no loss scaling on this path); one found in a
  reference-written file is ignored.

All writes go through a bounded retry (the reference retries 10 times because its blob storage fails occasionally).
"""
import json
import logging
import os
from typing import Callable, Dict, Union

import torch

LOGGER = logging.getLogger("xpretrain_amd")
MAX_IO_TRIALS = 10


def _map_tensors(state, fn: Callable[[torch.Tensor], torch.Tensor]):
    """apply ``fn`` to every tensor of a nested dict / list / tuple structure (anything else passes through)"""
    if torch.is_tensor(state):
        return fn(state)
    if isinstance(state, dict):
        return {k: _map_tensors(v, fn) for k, v in state.items()}
    if isinstance(state, (list, tuple)):
        return type(state)(_map_tensors(v, fn) for v in state)
    return state


def to_cpu_half(state):
    """storage form of the restorer: CPU tensors, fp32 narrowed to fp16 to halve the file"""
    return _map_tensors(state, lambda t: t.cpu().half() if t.dtype == torch.float32 else t.cpu())


def to_device_float(state, device):
    """inverse of ``to_cpu_half``: onto ``device``, fp16 widened to fp32 (masters and optimizer moments are fp32)"""
    return _map_tensors(state, lambda t: t.to(device).float() if t.dtype == torch.float16 else t.to(device))


def _retry(what: str, action: Callable[[], None]) -> bool:
    for trial in range(MAX_IO_TRIALS):
        try:
            action()
            return True
        except Exception as e:           # noqa: BLE001 -- the reference swallows and retries every failure
            LOGGER.warning("%s failed (trial %d of %d): %s", what, trial + 1, MAX_IO_TRIALS, e)
    return False


class ModelSaver:
    def __init__(self, output_dir):
        self.output_dir = output_dir

    def save(self, step, model, optimizer=None, prefix="model"):
        weights = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in model.state_dict().items()}

        def write():
            torch.save(weights, os.path.join(self.output_dir, f"{prefix}_step_{step}.pt"))
            if optimizer is not None:
                opt_state = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in optimizer.state_dict().items()}
                torch.save({"step": step, "optimizer": opt_state},
                           os.path.join(self.output_dir, f"{prefix}_step_{step}_train_state.pt"))
        return _retry(f"ModelSaver.save(step={step})", write)


def load_state_dict_with_mismatch(model, loaded_state_dict_or_path: Union[str, Dict[str, torch.Tensor]]):
    """In place.  Returns ``{'unexpected': [...], 'missing': [...], 'mismatched': [...]}`` (the reference only logs them)."""
    loaded = (torch.load(loaded_state_dict_or_path, map_location="cpu") if isinstance(loaded_state_dict_or_path, str)
              else loaded_state_dict_or_path)
    own = model.state_dict()
    usable, mismatched = {}, []
    for name, tensor in own.items():
        if name in loaded:
            if tuple(loaded[name].shape) == tuple(tensor.shape):
                usable[name] = loaded[name]
            else:
                mismatched.append(name)
    report = {"unexpected": sorted(set(loaded) - set(own)), "missing": sorted(set(own) - set(loaded)),
              "mismatched": sorted(mismatched)}
    for kind, names in report.items():
        LOGGER.info("load_state_dict_with_mismatch: %d %s keys %s", len(names), kind, names)
    model.load_state_dict(usable, strict=False)
    return report


class E2E_TrainingRestorer:
    """``opts`` needs ``output_dir``, ``save_steps_ratio`` and ``num_train_steps`` (``fp16`` is accepted and ignored)."""

    def __init__(self, opts, model, optimizer):
        out = opts.output_dir
        if os.path.exists(os.path.join(out, "log", "args.json")):          # keep the resumed run's arguments next to it
            with open(os.path.join(out, "log", "restore_args.json"), "w") as w:
                json.dump({k: v for k, v in vars(opts).items() if isinstance(v, (int, float, str, bool, list, dict, type(None)))},
                          w, indent=4)
        self.save_path = os.path.join(out, "restore.pt")
        self.backup_path = os.path.join(out, "restore_backup.pt")     # two generations, in case the newest is corrupt
        self.model, self.optimizer = model, optimizer
        self.save_steps = max(1, int(opts.save_steps_ratio * opts.num_train_steps))
        self.global_step = 0
        if os.path.exists(self.save_path) or os.path.exists(self.backup_path):
            _retry("E2E_TrainingRestorer.restore", self.restore)
        self.sync_ranks()

    def sync_ranks(self, src: int = 0):
        """Data-parallel runs: after the (per-rank) restore every rank takes rank ``src``'s step counter, parameters AND
        optimizer state (hvd.broadcast_parameters + hvd.broadcast_optimizer_state, run_pretrain.py:231-232) -- a rank that read
        the backup generation, or found no file, must not continue with different Adam moments.  No-op in a 1-rank run."""
        from .. import distributed as D
        if D.world_size() == 1:
            return
        import torch.distributed as dist
        box = [self.global_step]
        dist.broadcast_object_list(box, src)
        self.global_step = box[0]
        D.broadcast_parameters(self.model, src)
        D.broadcast_optimizer_state(self.optimizer, src)

    def step(self):
        self.global_step += 1
        if self.global_step % self.save_steps == 0:
            _retry(f"E2E_TrainingRestorer.save(step={self.global_step})", self.save)

    def save(self):
        checkpoint = {"global_step": self.global_step,
                      "model_state_dict": to_cpu_half(self.model.state_dict()),
                      "optim_state_dict": to_cpu_half(self.optimizer.state_dict())}
        if os.path.exists(self.save_path):
            os.replace(self.save_path, self.backup_path)
        torch.save(checkpoint, self.save_path)

    def restore(self):
        try:
            checkpoint = torch.load(self.save_path, map_location="cpu")
        except Exception:               # noqa: BLE001 -- fall back to the previous generation
            checkpoint = torch.load(self.backup_path, map_location="cpu")
        device = next(self.model.parameters()).device
        self.model.load_state_dict(to_device_float(checkpoint["model_state_dict"], device))
        self.optimizer.load_state_dict(to_device_float(checkpoint["optim_state_dict"], device))
        self.global_step = checkpoint["global_step"]
        LOGGER.info("resume training from step %d", self.global_step)
