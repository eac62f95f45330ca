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


def test_tower_segments_and_layer_groups_partition_the_parameters():
    """distributed.tower_segments / functional.layer_grad_groups on the real module tree: the two towers (each with its projection) are
    disjoint, together they hold every parameter but logit_scale, every encoder layer is one 16-parameter group inside ONE segment --
    so a GradBucketReducer built from them never mixes gradients of two streams in a bucket and hooks one parameter per layer."""
    import xpretrain_amd.functional as XF
    from xpretrain_amd import distributed as D
    from xpretrain_amd.modeling import VidCLIP
    m = VidCLIP(_Args(O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 3, 256, 120, 16, 64), temporal_size=2))
    segs = D.tower_segments(m)
    assert len(segs) == 2
    ids = [set(map(id, s)) for s in segs]
    assert not (ids[0] & ids[1])
    rest = [n for n, p in m.named_parameters() if id(p) not in ids[0] | ids[1]]
    assert rest == ["clipmodel.logit_scale"], rest
    names = {id(p): n for n, p in m.named_parameters()}
    assert all(names[i].startswith(("clipmodel.vision_model.", "clipmodel.visual_projection.")) for i in ids[0])
    assert all(names[i].startswith(("clipmodel.text_model.", "clipmodel.text_projection.")) for i in ids[1])
    groups = XF.layer_grad_groups(m)
    assert len(groups) == 2 + 3 and all(len(g) == 16 for g in groups)
    for g in groups:
        assert len({0 if id(p) in ids[0] else 1 for p in g}) == 1
    # single process: the reducer is inert, but its bucket plan is built all the same
    red = D.GradBucketReducer(m.parameters(), bucket_mb=0.5, layout_groups=groups, segments=segs)
    for b in red.buckets:
        assert len({(id(p) in ids[0], id(p) in ids[1]) for p in b["params"]}) == 1
    assert XF.second_chain_stream_mode() in ("side", "own")


def test_forward_hooks_on_encoder_layers_are_detected():
    """CLIPEncoder keeps the video tower on ONE chain while any encoder layer carries a forward hook (the hook would read the layer's
    output on the caller's stream before the second chain has written its half)."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.modeling.CLIP_ViP import _has_forward_hooks
    m = VidCLIP(_Args(O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 1, 256, 120, 16, 64), temporal_size=2))
    layers = m.clipmodel.vision_model.encoder.layers
    assert not _has_forward_hooks(layers)
    h = layers[1].register_forward_hook(lambda mod, inp, out: None)
    assert _has_forward_hooks(layers)
    h.remove()
    h = layers[0].register_forward_pre_hook(lambda mod, inp: None)
    assert _has_forward_hooks(layers)
    h.remove()
    assert not _has_forward_hooks(layers)
