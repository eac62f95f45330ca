"""Parameter grouping of the reference optimizer setup.

Mirrors ``src/optimization/utils.py``: ``build_e2e_optimizer_w_lr_mul`` :124-154 (four groups: {lr_mul prefix, rest} x
{decay, no-decay}; no-decay = names containing ``bias`` / ``LayerNorm.bias`` / ``LayerNorm.weight`` / ``logit_scale``)
and ``setup_e2e_optimizer`` :91-121 with ``optim == "adamw"`` (the only optimizer the CLIP-ViP configs use).
"""
from .adamw import AdamW

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight", "logit_scale")


def build_e2e_optimizer_w_lr_mul(model_param_optimizer, learning_rate, weight_decay, lr_mul=1, lr_mul_prefix=""):
    if lr_mul_prefix == "":
        param_optimizer, param_top = list(model_param_optimizer), []
    else:
        param_top = [(n, p) for n, p in model_param_optimizer if lr_mul_prefix in n and p.requires_grad]
        param_optimizer = [(n, p) for n, p in model_param_optimizer if lr_mul_prefix not in n and p.requires_grad]

    def nd(n):
        return any(x in n for x in NO_DECAY)

    return [
        {"params": [p for n, p in param_top if not nd(n)], "lr": lr_mul * learning_rate, "weight_decay": weight_decay},
        {"params": [p for n, p in param_top if nd(n)], "lr": lr_mul * learning_rate, "weight_decay": 0.0},
        {"params": [p for n, p in param_optimizer if not nd(n)], "weight_decay": weight_decay},
        {"params": [p for n, p in param_optimizer if nd(n)], "weight_decay": 0.0},
    ]


def setup_e2e_optimizer(model, opts):
    if getattr(opts, "optim", "adamw") != "adamw":
        raise ValueError("invalid optimizer")     # adam / adamax of the reference are not on the CLIP-ViP path
    groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), opts.learning_rate, opts.weight_decay,
                                          lr_mul=opts.lr_mul, lr_mul_prefix=opts.lr_mul_prefix)
    return AdamW(groups, lr=opts.learning_rate, betas=tuple(opts.betas))
