"""CPU: the drop-in surface -- class names, constructor arguments, state-dict key schema,
config plumbing, loud failure without a GPU."""
import pytest
import torch

from oracle import clipvip_oracle as O


class _Args:
    def __init__(self, cfg, temporal_size=12, add_cls_num=3, typ="ViP"):
        self.clip_config = cfg
        self.clip_weights = ""
        self.clip_vision_additional_config = dict(type=typ, temporal_size=temporal_size, if_use_temporal_embed=1,
                                                  logit_scale_init_value=4.6, add_cls_num=add_cls_num)


def test_state_dict_schema_vit_b16():
    from xpretrain_amd.modeling import VidCLIP
    with torch.device("meta"):
        m = VidCLIP(_Args(O.vit_b_config(16)))
    sd = m.state_dict()
    assert len(sd) == 402                                   # SURVEY.md §8b [probed on the reference]
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    assert shapes["clipmodel.logit_scale"] == ()
    assert shapes["clipmodel.vision_model.embeddings.added_cls"] == (3, 768)
    assert shapes["clipmodel.vision_model.embeddings.temporal_embedding"] == (1, 12, 768)
    assert shapes["clipmodel.vision_model.embeddings.position_ids"] == (1, 197)
    assert shapes["clipmodel.vision_model.embeddings.patch_embedding.weight"] == (768, 3, 16, 16)
    assert shapes["clipmodel.vision_model.pre_layrnorm.weight"] == (768,)          # (sic)
    assert shapes["clipmodel.text_model.embeddings.position_ids"] == (1, 77)
    assert shapes["clipmodel.text_model.embeddings.token_embedding.weight"] == (49408, 512)
    assert shapes["clipmodel.vision_model.encoder.layers.11.self_attn.q_proj.weight"] == (768, 768)
    assert shapes["clipmodel.text_model.encoder.layers.11.mlp.fc1.weight"] == (2048, 512)
    assert shapes["clipmodel.visual_projection.weight"] == (512, 768)
    assert shapes["clipmodel.text_projection.weight"] == (512, 512)
    assert sum(v.numel() for k, v in sd.items() if "position_ids" not in k) == 149632257


def test_reference_fixture_loads_strict(golden):
    from xpretrain_amd.modeling import VidCLIP
    fx = golden("tiny_e2e.pt")
    m = VidCLIP(_Args(fx["config"], fx["temporal_size"], fx["add_cls_num"]))
    m.load_state_dict(fx["state_dict"], strict=True)
    assert float(m.clipmodel.logit_scale) == pytest.approx(4.6)
    m.overload_logit_scale(1.0)
    assert float(m.clipmodel.logit_scale) == 1.0
    m.freeze_text_encoder(True)
    assert not m.clipmodel.text_projection.weight.requires_grad
    assert m.clipmodel.visual_projection.weight.requires_grad


def test_standalone_towers_construct():
    from xpretrain_amd.modeling import CLIPTextModel, CLIPVisionModel, load_clip_config
    cfg = load_clip_config(O.hf_config_dict(128, 2, 1, 256, 16, 32, 128, 2, 1, 256, 120, 16, 64))
    v = CLIPVisionModel(cfg.vision_config, dict(temporal_size=4, if_use_temporal_embed=1, add_cls_num=3))
    t = CLIPTextModel(cfg.text_config)
    assert "vision_model.pre_layrnorm.weight" in v.state_dict()
    assert "text_model.final_layer_norm.bias" in t.state_dict()


def test_non_vip_and_unknown_loss_fail_loudly():
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import build_loss_func
    with pytest.raises(NotImplementedError):
        VidCLIP(_Args(O.hf_config_dict(128, 2, 1, 256, 16, 32, 128, 2, 1, 256, 120, 16, 64), typ="ST"))
    assert build_loss_func({"loss_name": "NCELearnableTempLoss"}) is not None
    assert build_loss_func({"loss_name": "NCELearnableTempLoss_vsc_fc"}) is not None
    with pytest.raises(NotImplementedError):
        build_loss_func({"loss_name": "NCEHardNegLoss"})


def test_no_cpu_fallback():
    from xpretrain_amd.modeling import VidCLIP
    m = VidCLIP(_Args(O.hf_config_dict(128, 2, 1, 256, 16, 32, 128, 2, 1, 256, 120, 16, 64), temporal_size=2))
    video, ids, mask = O.synthetic_inputs(2, 2, 32, 8, vocab=120)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(video, ids, mask)


def test_output_objects_resolve_lazy_fields_everywhere():
    """`attentions` exists (None, as in the reference), dict(out) / out.copy() / pickle never leak a thunk, and a
    lazy field is evaluated under the grad mode of the forward that created it."""
    import pickle
    from xpretrain_amd.modeling.CLIP_ViP import BaseModelOutputWithPooling, _Lazy
    w = torch.ones(2, requires_grad=True)
    with torch.no_grad():
        lazy = _Lazy(lambda: w * 2)
    out = BaseModelOutputWithPooling(last_hidden_state=lazy, pooler_output=torch.zeros(1), hidden_states=None, attentions=None)
    assert out.attentions is None and out.hidden_states is None
    c = out.copy()
    assert torch.is_tensor(dict.__getitem__(c, "last_hidden_state"))
    assert not c.last_hidden_state.requires_grad                   # created under no_grad -> evaluated under no_grad
    d = dict(out.items())
    assert all(not isinstance(v, _Lazy) for v in d.values())
    r = pickle.loads(pickle.dumps(BaseModelOutputWithPooling(last_hidden_state=_Lazy(lambda: torch.ones(3)), pooler_output=None,
                                                             hidden_states=None, attentions=None)))
    assert torch.equal(r.last_hidden_state, torch.ones(3))
    assert out.to_tuple()[1] is out.pooler_output and len(out.to_tuple()) == 2
    with pytest.raises(AttributeError):
        out.no_such_field
