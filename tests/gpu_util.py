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


# ------------------------------------------------------------------------------------------------------------- tolerances
# ONE table for the end-to-end parity gates.  BASELINE.json:north_star states "within 1e-3 fp32 / 2e-2 bf16"
# for logits / loss on identical inputs; features are unit-norm, so the bf16 bound is applied ABSOLUTE to features and loss.
# Hidden states and gradients are compared as max|a-b| / max|b| (tensor scale):
#   *_emu  against the oracle that rounds to bf16 wherever the HIP path stores bf16 (values and activation gradients):
#          same arithmetic, so only accumulation order / fused-epilogue differences remain -- tight;
#   *_ref  against the UNMODIFIED reference's fp32 results (committed fixtures): includes the whole bf16 storage noise of a
#          12-layer network, measured 1-4e-2 on weight gradients and up to ~1e-1 on 1-D gradients of the 32-token text tower
#          (sums of a few hundred signed bf16-rounded rows that largely cancel).
TOL = {
    "features_abs": 2e-2,          # |vis - ref|, |txt - ref| (unit-norm features), bf16
    "cos_abs": 2e-2,               # |vis.txt^T - ref| (the logits before the learnable scale), bf16; measured ~5e-4
    "loss_abs": 2e-2,              # |loss - bf16-emulating oracle's loss|: two bf16 computations with the same storage points
    "loss_ref_abs": 6e-2,          # |loss - fp32 reference's loss| at logit scale e^4.6 ~ 100: a 3e-4 cosine error is a 3e-2
                                   # logit error.  The reference's OWN bf16 path (torch.autocast) moves the logits by 3.0e-2
                                   # and the loss by 0.5-3e-2 on the same inputs (tests/golden/full_cfg*.pt::ref_bf16)
    "fp32_abs": 1e-3,              # the same two in fp32 compute mode
    "hidden_emu": 1.2e-2,          # one layer (or a 2-layer model) vs the bf16-emulating oracle on the same input
    "hidden_emu_e2e": 3e-2,        # free-running 12-layer trajectories: two bf16 realisations decorrelate (measured 2e-2)
    "hidden_ref": 5e-2,            # hidden states vs reference fp32
    "grad_emu": 2e-2,              # teacher-forced per layer: dx and every parameter gradient vs the bf16-emulating oracle layer
                                   # fed the HIP path's own layer input and output gradient (measured <= 1.2e-2)
    "grad_emu_small": 1e-1,        # the same for the parameter gradients of the text tower at batch 2 (sums over 64 bf16-rounded
                                   # rows, in the last layers only the 2 pooled rows carry gradient: measured <= 6.7e-2; the
                                   # causal-attention and small-GEMM kernels themselves are held to 1e-2..2e-2 against fp64 in
                                   # tests/test_attention_gpu.py / test_gemm_gpu.py)
    "grad_ref_2d": 8e-2,           # weight gradients vs reference fp32
    "grad_ref_1d": 2e-1,           # 1-D gradients vs reference fp32 (worst: text-tower LayerNorm / bias gradients at batch 2,
                                   # sums over 64 rows: 1.5e-1 measured)
}


class ModelArgs:
    """the ``args`` object VidCLIP.__init__ reads (modeling/VidCLIP.py:9-27)"""

    def __init__(self, cfg, temporal_size, add_cls_num=3):
        self.clip_config = cfg
        self.clip_weights = ""
        self.clip_vision_additional_config = dict(type="ViP", temporal_size=temporal_size, if_use_temporal_embed=1,
                                                  logit_scale_init_value=4.6, add_cls_num=add_cls_num)


def seeded_model(cfgd, temporal_size, seed=1234, perturb_seed=99):
    """VidCLIP on the CPU with reproducible weights: reference init under ``manual_seed(seed)``, then biases, LayerNorm
    affine parameters and the temporal table -- all (0 | 1) at init (CLIP_ViP.py:166,481-522) -- perturbed from a second
    seeded generator so that every term of the forward and backward is exercised.  tests/golden/make_golden.py builds the
    SAME weights in the build container, loads them into the unmodified reference and stores its outputs; the GPU box
    rebuilds them from the seeds (CPU RNG streams are machine-independent for a fixed torch version)."""
    from xpretrain_amd.modeling import VidCLIP
    torch.manual_seed(seed)
    model = VidCLIP(ModelArgs(cfgd, temporal_size))
    g = torch.Generator().manual_seed(perturb_seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("logit_scale"):
                continue
            if n.endswith(".bias") or "temporal_embedding" in n:
                p.add_(0.02 * torch.randn(p.shape, generator=g))
            elif "layer_norm" in n or "layrnorm" in n or "layernorm" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return model


def sample_rows(S, M, n=28, seed=5):
    """proxy rows + n seeded token rows of a sample: the hidden-state rows the full-size fixtures keep"""
    g = torch.Generator().manual_seed(seed)
    return torch.cat([torch.arange(M), M + torch.randperm(S - M, generator=g)[:n].sort().values])
