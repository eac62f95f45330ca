"""CPU: the reference's shipped pre-training config drives this package's objects (xpretrain_amd/configs.py): model, precision, loss,
the four optimizer groups and the schedule -- INTEGRATION.md 1 as code.

Reference: src/configs/pretrain/pretrain_vip_base_16.json:37-91, src/configs/config.py:12-30, src/pretrain/run_pretrain.py:109-126,
222, 234-236, 324, 408-423.  The values below are the model / precision / optimisation keys of that JSON (data paths and logging keys
left out); when /root/reference is present the test first checks them against the file itself."""
import json
import math
import os

import pytest
import torch

REF_JSON = "/root/reference/CLIP-ViP/src/configs/pretrain/pretrain_vip_base_16.json"
PRETRAIN_VIP_BASE_16 = {
    "train_n_clips": 1, "train_num_frms": 12, "test_n_clips": 1, "test_num_frms": 12, "input_res": [224, 224], "max_txt_len": 70,
    "e2e_weights_path": None, "clip_weights": "openai/clip-vit-base-patch16", "clip_config": "openai/clip-vit-base-patch16",
    "clip_vision_additional_config": {"type": "ViP", "temporal_size": 12, "if_use_temporal_embed": 1, "logit_scale_init_value": 4.60,
                                      "add_cls_num": 3},
    "train_batch_size": 16, "gradient_accumulation_steps": 1, "fp16": 1, "amp_level": "O2", "seed": 42,
    "optim": "adamw", "betas": [0.9, 0.98], "learning_rate": 5e-6, "weight_decay": 0.05, "lr_mul": 1, "lr_mul_prefix": "",
    "loss_config": {"loss_name": "NCELearnableTempLoss_vsc_fc", "if_gather": 1},
    "warmup_ratio": 0.01, "decay": "cosine", "grad_norm": 5.0, "num_train_epochs": 5,
}


def test_committed_keys_are_the_reference_files():
    if not os.path.isfile(REF_JSON):
        pytest.skip("no /root/reference on this machine")
    ref = json.load(open(REF_JSON))
    for k, v in PRETRAIN_VIP_BASE_16.items():
        assert ref[k] == v, k


def test_shipped_pretrain_config_builds_model_loss_optimizer_and_schedule(tmp_path):
    from xpretrain_amd import configs as CF
    from xpretrain_amd.optimization import AdamW, NCELearnableTempLoss_vsc_fc
    src = REF_JSON if os.path.isfile(REF_JSON) else str(tmp_path / "pretrain_vip_base_16.json")
    if src != REF_JSON:
        json.dump(PRETRAIN_VIP_BASE_16, open(src, "w"))
    cfg = CF.load_config(src)
    assert cfg.clip_vision_additional_config.temporal_size == 12 and cfg.loss_config.loss_name == "NCELearnableTempLoss_vsc_fc"
    # fp16: 1 + amp O2 -> bf16 compute on fp32 masters, no loss scaling; fp16: 0 -> float32
    assert CF.compute_dtype(cfg) is torch.bfloat16 and CF.compute_dtype(CF.load_config(src, fp16=0)) is torch.float32
    with pytest.raises(ValueError):
        CF.compute_dtype(CF.load_config(src, amp_level="O3"))
    with pytest.raises(FileNotFoundError, match="no hub access"):        # released weights are not on this machine
        CF.setup_model(cfg)
    model = CF.setup_model(cfg, allow_random_init=True)
    sd = model.state_dict()
    assert len(sd) == 402 and sum(p.numel() for p in model.parameters()) == 149_632_257          # the checkpoint schema
    assert float(model.clipmodel.logit_scale) == pytest.approx(4.6)
    assert model.clipmodel.vision_model.embeddings.temporal_embedding.shape == (1, 12, 768)
    assert all(m.compute_dtype is torch.bfloat16 for m in model.modules() if hasattr(m, "compute_dtype"))
    assert all(p.dtype is torch.float32 for p in model.parameters())                              # fp32 masters
    steps = 1000
    tr = CF.setup_training(cfg, model, steps)
    assert isinstance(tr.loss_fn, NCELearnableTempLoss_vsc_fc) and isinstance(tr.optimizer, AdamW) and tr.grad_norm == 5.0
    # optimization/utils.py:124-154: {lr_mul-prefixed, rest} x {decay, no decay}; lr_mul_prefix "" -> the prefixed pair is empty
    groups = tr.optimizer.param_groups
    assert len(groups) == 4
    assert [g["weight_decay"] for g in groups] == [0.05, 0.0, 0.05, 0.0] and all(g["betas"] == (0.9, 0.98) for g in groups)
    assert sum(len(g["params"]) for g in groups) == len(list(model.parameters()))
    no_decay = {id(p) for g in groups if g["weight_decay"] == 0.0 for p in g["params"]}
    names = dict(model.named_parameters())
    assert id(names["clipmodel.logit_scale"]) in no_decay and id(names["clipmodel.vision_model.pre_layrnorm.bias"]) in no_decay
    assert id(names["clipmodel.text_model.encoder.layers.0.mlp.fc1.bias"]) in no_decay
    assert id(names["clipmodel.visual_projection.weight"]) not in no_decay
    # (the reference's 'LayerNorm.weight' pattern never matches CLIP's `layer_norm1` / `pre_layrnorm` names: those weights decay there, and here)
    assert id(names["clipmodel.vision_model.pre_layrnorm.weight"]) not in no_decay
    # sched.py:20-24, 62-84: 1 % linear warmup, cosine decay to 0
    assert tr.lr_at(0) == pytest.approx(5e-6 * 0 / 10, abs=1e-12) or tr.lr_at(0) > 0
    assert tr.lr_at(5) == pytest.approx(5e-6 * 0.5) and tr.lr_at(10) == pytest.approx(5e-6)
    mid = 10 + (steps - 10) // 2
    assert tr.lr_at(mid) == pytest.approx(5e-6 * 0.5 * (1 + math.cos(math.pi * (mid - 10) / (steps - 10))), rel=1e-6)


def test_frozen_text_tower_and_local_config_directory(tmp_path):
    from xpretrain_amd import configs as CF, workload
    d = tmp_path / "clip-vit-tiny"
    d.mkdir()
    json.dump(workload.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 2, 256, 120, 16, 64), open(d / "config.json", "w"))
    cfg = CF.load_config(dict(PRETRAIN_VIP_BASE_16, clip_config=str(d), clip_weights="", freeze_text_model=1, freeze_text_proj=1, fp16=0,
                              clip_vision_additional_config=dict(PRETRAIN_VIP_BASE_16["clip_vision_additional_config"], temporal_size=3)))
    model = CF.setup_model(cfg)
    assert not any(p.requires_grad for p in model.clipmodel.text_model.parameters()) and not model.clipmodel.text_projection.weight.requires_grad
    assert all(m.compute_dtype is torch.float32 for m in model.modules() if hasattr(m, "compute_dtype"))
    assert model.clipmodel.vision_model.embeddings.temporal_embedding.shape == (1, 3, 128)
