"""Shared helpers for the -m gpu parity tests."""
import json
import os

import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def dump(name, obj):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f)


def maxrel(a: torch.Tensor, b: torch.Tensor) -> float:
    """max|a-b| / max|b| (error relative to the tensor scale)."""
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def report(tag, a, b, tol, log=None):
    e = maxrel(a, b)
    line = f"{tag}: maxrel={e:.3e} tol={tol:.1e} {'OK' if e <= tol else 'FAIL'}"
    print(line)
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "parity_log.txt"), "a") as f:
        f.write(line + "\n")
    return e


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(t.dtype)
