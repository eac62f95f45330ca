"""CPU: checkpoint I/O with the reference's on-disk conventions (xpretrain_amd/utils/load_save.py; reference
src/utils/load_save.py:38-63, 86-115, 159-192, 260-327): key schema, fp16-stored checkpoints, shape-mismatch tolerance,
restore.pt round trip with backup generation."""
import os
import types

import pytest
import torch

from oracle import clipvip_oracle as O
from tests.test_model_cpu import _Args
from xpretrain_amd.utils.load_save import (E2E_TrainingRestorer, ModelSaver, load_state_dict_with_mismatch, to_cpu_half,
                                           to_device_float)

CFG = O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 2, 256, 120, 16, 64)


def _model(temporal_size=4, seed=0):
    from xpretrain_amd.modeling import VidCLIP
    torch.manual_seed(seed)
    return VidCLIP(_Args(CFG, temporal_size=temporal_size))


def test_model_saver_writes_the_plain_reference_schema(tmp_path):
    m = _model()
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    assert ModelSaver(str(tmp_path)).save(7, m, optimizer=opt)
    sd = torch.load(tmp_path / "model_step_7.pt")
    assert list(sd.keys()) == list(m.state_dict().keys()) and all(k.startswith("clipmodel.") for k in sd)
    assert all(v.device.type == "cpu" for v in sd.values())
    assert sd["clipmodel.vision_model.pre_layrnorm.weight"].dtype == torch.float32        # full precision, reference typo kept
    ts = torch.load(tmp_path / "model_step_7_train_state.pt")
    assert ts["step"] == 7 and set(ts["optimizer"]) == {"state", "param_groups"}
    m2 = _model(seed=1)
    m2.load_state_dict(sd, strict=True)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_fp16_stored_checkpoint_and_shape_mismatch(tmp_path):
    src = _model(temporal_size=4)
    half = {k: (v.half() if v.is_floating_point() else v) for k, v in src.state_dict().items()}     # released .pt files
    half["clipmodel.some_task_head.weight"] = torch.zeros(3, 3)
    del half["clipmodel.text_projection.weight"]
    path = str(tmp_path / "released.pt")
    torch.save(half, path)
    dst = _model(temporal_size=8, seed=3)               # temporal_embedding [1,8,D] vs [1,4,D] in the file
    before_te = dst.state_dict()["clipmodel.vision_model.embeddings.temporal_embedding"].clone()
    before_tp = dst.state_dict()["clipmodel.text_projection.weight"].clone()
    rep = load_state_dict_with_mismatch(dst, path)
    assert rep == {"unexpected": ["clipmodel.some_task_head.weight"], "missing": ["clipmodel.text_projection.weight"],
                   "mismatched": ["clipmodel.vision_model.embeddings.temporal_embedding"]}
    got = dst.state_dict()
    assert torch.equal(got["clipmodel.vision_model.embeddings.temporal_embedding"], before_te)
    assert torch.equal(got["clipmodel.text_projection.weight"], before_tp)
    w = "clipmodel.vision_model.encoder.layers.1.mlp.fc1.weight"
    assert got[w].dtype == torch.float32 and torch.equal(got[w], src.state_dict()[w].half().float())
    assert torch.equal(got["clipmodel.text_model.embeddings.position_ids"], src.state_dict()["clipmodel.text_model.embeddings.position_ids"])


def test_half_storage_helpers():
    state = {"a": torch.randn(3), "b": [torch.arange(3), (torch.randn(2, 2), 5)], "c": "x"}
    packed = to_cpu_half(state)
    assert packed["a"].dtype == torch.float16 and packed["b"][0].dtype == torch.int64 and packed["b"][1][1] == 5 and packed["c"] == "x"
    back = to_device_float(packed, "cpu")
    assert back["a"].dtype == torch.float32 and torch.equal(back["a"], state["a"].half().float())
    assert isinstance(back["b"][1], tuple)


def test_restorer_round_trip_with_backup_generation(tmp_path):
    opts = types.SimpleNamespace(output_dir=str(tmp_path), save_steps_ratio=0.25, num_train_steps=8, fp16=1)
    os.makedirs(tmp_path / "log")
    (tmp_path / "log" / "args.json").write_text("{}")
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    r = E2E_TrainingRestorer(opts, net, opt)
    assert r.global_step == 0 and r.save_steps == 2
    torch.manual_seed(0)
    for _ in range(5):                                   # saves at steps 2 and 4 -> restore.pt (4) + restore_backup.pt (2)
        net(torch.randn(4, 6)).pow(2).sum().backward()
        opt.step(); opt.zero_grad(); r.step()
    assert os.path.exists(r.save_path) and os.path.exists(r.backup_path) and os.path.exists(tmp_path / "log" / "restore_args.json")
    ck = torch.load(r.save_path)
    assert ck["global_step"] == 4 and ck["model_state_dict"]["0.weight"].dtype == torch.float16 and "amp_state_dict" not in ck
    assert torch.load(r.backup_path)["global_step"] == 2
    net2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5))
    opt2 = torch.optim.Adam(net2.parameters(), lr=1e-2)
    r2 = E2E_TrainingRestorer(opts, net2, opt2)          # resumes from restore.pt
    assert r2.global_step == 4
    assert torch.equal(net2[0].weight, ck["model_state_dict"]["0.weight"].float())
    st = opt2.state_dict()["state"][0]
    assert st["exp_avg"].dtype == torch.float32 and int(st["step"]) == 4
    os.remove(r.save_path)                               # newest generation lost -> the backup is used
    (tmp_path / "restore.pt").write_bytes(b"corrupt")
    r3 = E2E_TrainingRestorer(opts, torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5)),
                              torch.optim.Adam(net2.parameters(), lr=1e-2))
    assert r3.global_step == 2
