"""Shared helpers for the -m gpu parity tests."""
import json
import os

import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def dump(name, obj):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f)


def maxrel(a: torch.Tensor, b: torch.Tensor, scale_floor: float = 1e-30) -> float:
    """max|a-b| / max(max|b|, scale_floor) (error relative to the tensor scale)."""
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(scale_floor)).item()


def report(tag, a, b, tol, log=None, scale_floor: float = 1e-30):
    e = maxrel(a, b, scale_floor)
    line = f"{tag}: maxrel={e:.3e} tol={tol:.1e} {'OK' if e <= tol else 'FAIL'}"
    print(line)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_log.txt"), "a") as f:
        f.write(line + "\n")
    return e


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(t.dtype)
