#!/usr/bin/env python
"""CPU experiment (oracle with bf16 storage emulation, tests/golden/full_cfg*.pt as the fp32 truth): how much of the loss deviation of
a bf16-storage implementation comes from rounding the RESIDUAL STREAM, and how much of that from the M proxy rows alone (the pooled
feature is token 0 of the last layer; frame-token errors reach it only through attention averages).  VERDICT r2 #3 / "weak" #1.
Usage: python tools/residual_precision_experiment.py [full_cfg2.pt ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import clipvip_oracle as O  # noqa: E402
from tests.gpu_util import seeded_model  # noqa: E402

MODE = {"m": "all"}
orig_layer = O.encoder_layer


def layer(x, sd, pfx, heads, size, pad_mask):
    R = O.ROUND
    video = size is not None

    def rnd(t):
        if MODE["m"] == "all" or not video:
            return R(t)
        if MODE["m"] == "resid_fp32":
            return t
        Mp = size[0]                                 # proxy rows keep fp32, the rest is rounded
        return torch.cat([t[:, :Mp], R(t[:, Mp:])], dim=1)
    h = O.layer_norm(x, sd[pfx + "layer_norm1.weight"], sd[pfx + "layer_norm1.bias"])
    x = rnd(x + O.attention_block(h, sd, pfx + "self_attn.", heads, size, pad_mask))
    h = O.layer_norm(x, sd[pfx + "layer_norm2.weight"], sd[pfx + "layer_norm2.bias"])
    h = O.linear(R(O.quick_gelu(O.linear(h, sd, pfx + "mlp.fc1"))), sd, pfx + "mlp.fc2")
    return rnd(x + h)


O.encoder_layer = layer
for name in sys.argv[1:] or ["full_cfg2.pt"]:
    fx = torch.load(os.path.join(ROOT, "tests", "golden", name), map_location="cpu", weights_only=False)
    cfgd = O.vit_b_config(fx["patch"], fx["res"])
    model = seeded_model(cfgd, fx["temporal_size"])
    sd = {k: v.detach() for k, v in O.strip_prefix(model.state_dict()).items()}
    video, ids, mask = O.synthetic_inputs(fx["B"], fx["frames"], fx["res"], fx["txt_len"])
    cfg = O.OracleCfg.from_hf_dict(cfgd, temporal_size=fx["temporal_size"])
    print(f"{name}: reference fp32 loss {fx['loss'].item():.5f}; its own bf16 autocast {fx['ref_bf16']['loss']:.5f}")
    for mode in ("all", "proxy_fp32", "resid_fp32"):
        MODE["m"] = mode
        O.ROUND.dtype = torch.bfloat16
        try:
            with torch.no_grad():
                loss, vis, txt = O.full_step(video, ids, mask, sd, cfg)
        finally:
            O.ROUND.dtype = None
        dv = (vis - fx["vis_features"]).abs().max().item()
        dcos = (vis @ txt.t() - fx["vis_features"] @ fx["text_features"].t()).abs().max().item()
        print(f"  bf16 storage, residual stream {mode:10s}: loss {loss.item():.5f}  |d loss| {abs(loss.item() - fx['loss'].item()):.2e} "
              f"({100 * abs(loss.item() - fx['loss'].item()) / fx['loss'].item():.2f} %)  |d vis| {dv:.2e}  |d cos| {dcos:.2e}")
