"""Contrastive loss surface (reference: src/optimization/loss.py).

``NCELearnableTempLoss`` (loss.py:126-141) is the hot-path loss: fused HIP kernel computing
exp(temp) * vis @ text^T, both cross-entropies and all gradients in one call
(csrc/loss.hip).  ``build_loss_func(cfg)`` keeps the reference's factory signature (loss.py:326-328).
"""
from __future__ import annotations

from torch import nn

from .. import functional as XF


class NCELearnableTempLoss(nn.Module):
    """loss = CE(exp(temp) * V T^T, diag) + CE(exp(temp) * T V^T, diag); `temp` is the LOG-scale parameter."""

    def __init__(self, cfg=None):
        super().__init__()

    def forward(self, vis_feat, text_feat, temp):
        return XF.NCELossFn.apply(vis_feat, text_feat, temp)


class NCELearnableTempLoss_vsc_fc(nn.Module):
    """Pre-training default (loss.py:288-324, pretrain_vip_base_16.json:75): video-(subtitle, caption) and
    frame-caption contrast, six cross-entropy terms; same call signature as the reference."""

    def __init__(self, cfg=None):
        super().__init__()

    def forward(self, vis_feat, text_feat, img_feat, cap_feat, temp):
        assert text_feat.shape[0] == cap_feat.shape[0]                       # loss.py:297
        return XF.VscFcLossFn.apply(vis_feat, text_feat, img_feat, cap_feat, temp)


_LOSSES = {"NCELearnableTempLoss": NCELearnableTempLoss, "NCELearnableTempLoss_vsc_fc": NCELearnableTempLoss_vsc_fc}


def build_loss_func(cfg):
    name = cfg["loss_name"] if isinstance(cfg, dict) else cfg.loss_name
    if name not in _LOSSES:
        raise NotImplementedError(f"loss {name!r} is not on the CLIP-ViP path; available on the HIP path: {sorted(_LOSSES)}")
    return _LOSSES[name](cfg)
