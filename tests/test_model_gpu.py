"""GPU: end-to-end parity of the HIP path (VidCLIP.forward -> NCELearnableTempLoss -> backward) against
(1) the reference-generated fixture tests/golden/tiny_e2e.pt (fp32 outputs of the UNMODIFIED reference) and
(2) the CPU oracle at BASELINE cfg #1's architecture (ViT-B/32, T=2, Lt=16), bf16 tolerance 2e-2 as
BASELINE.json:north_star states (on unit-norm features / loss / max-norm-relative gradients)."""
import pytest
import torch

from oracle import clipvip_oracle as O
from tests.gpu_util import TOL, loss_gate, report

pytestmark = pytest.mark.gpu


class _Args:
    def __init__(self, cfg, temporal_size, add_cls_num=3):
        self.clip_config = cfg
        self.clip_weights = ""
        self.clip_vision_additional_config = dict(type="ViP", temporal_size=temporal_size, if_use_temporal_embed=1,
                                                  logit_scale_init_value=4.6, add_cls_num=add_cls_num)


def test_tiny_e2e_against_reference_fixture(golden):
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import NCELearnableTempLoss
    fx = golden("tiny_e2e.pt")
    model = VidCLIP(_Args(fx["config"], fx["temporal_size"], fx["add_cls_num"]))
    model.load_state_dict(fx["state_dict"], strict=True)
    model.cuda().train()
    # The fixture model has 3x-widened weights (sharp softmaxes, |activations| ~ 5) -- hostile to bf16 storage.
    # So each hidden state is checked twice: against the oracle EMULATING bf16 storage at the kernels' rounding
    # points (tight: tells a kernel bug from bf16 noise) and against the reference's fp32 values (loose).
    cfg = O.OracleCfg.from_hf_dict(fx["config"], add_cls_num=fx["add_cls_num"], temporal_size=fx["temporal_size"])
    sd = O.strip_prefix(fx["state_dict"])
    O.ROUND.dtype = torch.bfloat16
    try:
        emu_v, emu_t = [], []
        _, emu_vp = O.vision_tower(fx["video"], sd, cfg, collect=emu_v)
        _, emu_tp = O.text_tower(fx["ids"], fx["mask"], sd, cfg, collect=emu_t)
    finally:
        O.ROUND.dtype = None
    vo = model.clipmodel.vision_model(pixel_values=fx["video"].cuda(), output_hidden_states=True)
    for i, (a, b, e) in enumerate(zip(vo["hidden_states"], fx["vision_hidden"], emu_v[1:])):
        assert report(f"tiny vision hidden[{i}] vs bf16-emulating oracle", a, e, 1.2e-2) <= 1.2e-2
        assert report(f"tiny vision hidden[{i}] vs reference fp32", a, b, 5e-2) <= 5e-2
    assert report("tiny vision pooled vs emu", vo["pooler_output"], emu_vp, 1.5e-2) <= 1.5e-2
    to = model.clipmodel.text_model(input_ids=fx["ids"].cuda(), attention_mask=fx["mask"].cuda(), output_hidden_states=True)
    for i, (a, b, e) in enumerate(zip(to["hidden_states"], fx["text_hidden"], emu_t)):
        assert report(f"tiny text hidden[{i}] vs bf16-emulating oracle", a, e, 1.2e-2) <= 1.2e-2
        assert report(f"tiny text hidden[{i}] vs reference fp32", a, b, 5e-2) <= 5e-2
    assert report("tiny text pooled vs emu", to["pooler_output"], emu_tp, 1.5e-2) <= 1.5e-2

    out = model(fx["video"].cuda(), fx["ids"].cuda(), fx["mask"].cuda())
    dv = (out["vis_features"].cpu() - fx["vis_features"]).abs().max().item()
    dt = (out["text_features"].cpu() - fx["text_features"]).abs().max().item()
    loss = NCELearnableTempLoss()(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
    print(f"tiny: dvis {dv:.3e} dtxt {dt:.3e} loss {loss.item():.5f} ref {fx['loss'].item():.5f}")
    assert dv < 8e-3 and dt < 5e-3                           # measured 5.1e-3 / 3.0e-3 (3x-widened weights)
    assert loss_gate(loss.item(), fx["loss"].item(), 5e-2)  # measured 2.3e-2 on a loss of 25.5 (0.1 %)
    loss.backward()
    bad = []
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        ref = fx["grads"][name]
        e = report(f"tiny grad {name}", p.grad, ref, 1.5e-1) if ref.abs().max() > 1e-4 else 0.0
        if e > 1.5e-1:     # loose: bf16 through the widened-weight fixture; the realistic-init test below is the 2e-2 gate
            bad.append((name, e))
    assert not bad, bad


def _oracle_step(model, video, ids, mask, cfg, emulate):
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in O.strip_prefix(model.state_dict()).items()}
    O.ROUND.dtype, O.ROUND.grads = (torch.bfloat16, True) if emulate else (None, False)
    try:
        loss, vis, txt = O.full_step(video, ids, mask, sd, cfg)
        loss.backward()
    finally:
        O.ROUND.dtype, O.ROUND.grads = None, False
    return loss.detach(), vis.detach(), txt.detach(), {k: v.grad for k, v in sd.items() if v.grad is not None}


def test_cfg1_architecture_against_oracle():
    """BASELINE config #1: ViT-B/32, 2 frames 224^2, 16 text tokens, batch 2 -- features, loss and EVERY parameter gradient
    against the oracle in fp32 and with bf16 storage emulation (tolerances: tests/gpu_util.py::TOL)."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import NCELearnableTempLoss
    torch.manual_seed(1234)
    cfgd = O.vit_b_config(patch=32)
    model = VidCLIP(_Args(cfgd, 12))
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
    video, ids, mask = O.synthetic_inputs(2, 2, 224, 16)
    cfg = O.OracleCfg.from_hf_dict(cfgd)
    ref_loss, ref_vis, ref_txt, ref_g = _oracle_step(model, video, ids, mask, cfg, emulate=False)
    emu_loss, emu_vis, emu_txt, emu_g = _oracle_step(model, video, ids, mask, cfg, emulate=True)
    model.cuda().train()
    out = model(video.cuda(), ids.cuda(), mask.cuda())
    loss = NCELearnableTempLoss()(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
    loss.backward()
    dv = (out["vis_features"].cpu() - ref_vis).abs().max().item()
    dt = (out["text_features"].cpu() - ref_txt).abs().max().item()
    print(f"cfg1: loss {loss.item():.6f} oracle fp32 {ref_loss.item():.6f} emulation {emu_loss.item():.6f} dvis {dv:.2e} dtxt {dt:.2e}")
    assert dv <= TOL["features_abs_full"] and dt <= TOL["features_abs_full"]
    assert loss_gate(loss.item(), ref_loss.item(), TOL["loss_ref_abs"])
    # The free-running emulation is reported, not gated: two bf16 realisations of a 12-layer network are as far from each
    # other as from fp32; the tight emulation gates are the teacher-forced per-layer ones of test_fullsize_parity_gpu.py.
    worst = {"grad_ref_2d": 0.0, "grad_ref_1d": 0.0}
    info = {"grad_emu": 0.0}
    for name, p in model.named_parameters():
        key = name[len("clipmodel."):]
        assert p.grad is not None and key in ref_g, name
        if ref_g[key].abs().max() <= 1e-5:
            continue
        k_ref = "grad_ref_2d" if p.dim() >= 2 else "grad_ref_1d"
        worst[k_ref] = max(worst[k_ref], report(f"cfg1 grad {name} vs fp32 oracle", p.grad, ref_g[key], TOL[k_ref]))
        info["grad_emu"] = max(info["grad_emu"], report(f"cfg1 grad {name} vs free-running emulation (info)", p.grad, emu_g[key], 1.0))
    print("cfg1 worst gradient errors", worst, info)
    assert all(worst[k] <= TOL[k] for k in worst), worst


def test_second_pass_T1_interpolated_temporal_embedding():
    """VidCLIP's image/caption pass runs the video tower at T=1 (VidCLIP.py:70-79): exercises the linear
    interpolation of temporal_embedding 12 -> 1 and its gradient."""
    from xpretrain_amd.modeling import VidCLIP
    torch.manual_seed(7)
    cfgd = O.hf_config_dict(128, 2, 1, 256, 16, 32, 128, 2, 1, 256, 120, 16, 64)
    model = VidCLIP(_Args(cfgd, 12))
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.5)
    video, ids, mask = O.synthetic_inputs(3, 2, 32, 8, vocab=120)
    image = video[:, :1].contiguous()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in O.strip_prefix(model.state_dict()).items()}
    cfg = O.OracleCfg.from_hf_dict(cfgd)
    rv, _ = O.clip_features(image, ids, mask, sd, cfg)
    rv.sum().backward()
    model.cuda()
    out = model(video.cuda(), ids.cuda(), mask.cuda(), image=image.cuda(), caption_ids=ids[:, None].cuda(),
                caption_masks=mask[:, None].cuda())
    assert (out["img_features"].cpu() - rv).abs().max() < 2e-2
    out["img_features"].sum().backward()
    g = model.clipmodel.vision_model.embeddings.temporal_embedding.grad.cpu()
    gr = sd["vision_model.embeddings.temporal_embedding"].grad
    assert report("temporal_embedding grad (T=1 interp)", g, gr, 6e-2) <= 6e-2


def test_weight_cache_follows_optimizer_steps():
    """bf16 weight copies must track the fp32 masters across (fused) optimizer steps: two SGD-like steps change the
    features; a stale cache would reproduce the first output exactly."""
    from xpretrain_amd.modeling import VidCLIP
    torch.manual_seed(3)
    cfgd = O.hf_config_dict(128, 2, 1, 256, 16, 32, 128, 2, 1, 256, 120, 16, 64)
    model = VidCLIP(_Args(cfgd, 2)).cuda()
    video, ids, mask = O.synthetic_inputs(2, 2, 32, 8, vocab=120)
    video, ids, mask = video.cuda(), ids.cuda(), mask.cuda()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=True)
    f0 = model(video, ids, mask)["vis_features"].detach().clone()
    model(video, ids, mask)["vis_features"].sum().backward()
    opt.step()
    f1 = model(video, ids, mask)["vis_features"].detach()
    assert (f1 - f0).abs().max().item() > 1e-3
    # and the refreshed copy equals the master weights: compare against the oracle on the updated state dict
    sd = {k: v.detach().cpu() for k, v in O.strip_prefix(model.state_dict()).items()}
    ref, _ = O.clip_features(video.cpu(), ids.cpu(), mask.cpu(), sd, O.OracleCfg.from_hf_dict(cfgd, temporal_size=2))
    assert (f1.cpu() - ref).abs().max().item() < 2e-2


def test_pretrain_step_dual_pass_vsc_fc_against_oracle():
    """the pre-training step of run_pretrain -- video+subtitle pass, middle-frame image + caption pass at
    T=1 (VidCLIP.py:70-79), NCELearnableTempLoss_vsc_fc (loss.py:288-324) -- forward, loss and EVERY parameter gradient
    (both passes accumulate into the shared weights) against the fp32 oracle."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import build_loss_func
    torch.manual_seed(11)
    cfgd = O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 2, 256, 120, 16, 64)
    model = VidCLIP(_Args(cfgd, 4))
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.1)
    B = 4
    video, ids, mask = O.synthetic_inputs(B, 4, 32, 8, vocab=120)
    _, cap_ids, cap_mask = O.synthetic_inputs(B, 1, 32, 8, vocab=120, seed=99)
    image = video[:, 1:2].contiguous()
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in O.strip_prefix(model.state_dict()).items()}
    cfg = O.OracleCfg.from_hf_dict(cfgd, temporal_size=4)
    rv, rt = O.clip_features(video, ids, mask, sd, cfg)
    ri, rc = O.clip_features(image, cap_ids, cap_mask, sd, cfg)
    ref_loss = O.nce_vsc_fc_loss(rv, rt, ri, rc, sd["logit_scale"])
    ref_loss.backward()
    model.cuda().train()
    out = model(video.cuda(), ids.cuda(), mask.cuda(), image=image.cuda(), caption_ids=cap_ids[:, None].cuda(),
                caption_masks=cap_mask[:, None].cuda())
    loss = build_loss_func({"loss_name": "NCELearnableTempLoss_vsc_fc"})(
        out["vis_features"], out["text_features"], out["img_features"], out["cap_features"], model.clipmodel.logit_scale)
    loss.backward()
    for k, r in (("vis_features", rv), ("text_features", rt), ("img_features", ri), ("cap_features", rc)):
        assert (out[k].detach().cpu() - r).abs().max().item() < 2e-2, k
    print(f"pretrain step: loss {loss.item():.5f} oracle {ref_loss.item():.5f}")
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * max(1.0, abs(ref_loss.item()))
    worst = 0.0
    for name, p in model.named_parameters():
        ref = sd[name[len("clipmodel."):]].grad
        assert p.grad is not None and ref is not None, name
        if ref.abs().max() > 1e-5:
            worst = max(worst, report(f"pretrain-step grad {name}", p.grad, ref, 8e-2))
    assert worst <= 8e-2


def test_output_fields_frozen_text_tower_and_stale_weight_guard():
    """(1) every output field of the reference is there (lazily computed ones too) and an unknown name raises;
    (2) freeze_text_encoder (VidCLIP.py:96-103): the text tower's parameter-gradient kernels are skipped, the video tower's
    gradients are bit-identical to the unfrozen run; (3) forward -> optimizer.step() -> backward fails loudly: the bf16
    weight copies saved for backward were rewritten by the optimizer kernel."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import AdamW, NCELearnableTempLoss
    torch.manual_seed(5)
    cfgd = O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 2, 256, 120, 16, 64)
    model = VidCLIP(_Args(cfgd, 2)).cuda().train()
    video, ids, mask = (t.cuda() for t in O.synthetic_inputs(3, 2, 32, 8, vocab=120))
    sd = {k: v.detach().cpu() for k, v in O.strip_prefix(model.state_dict()).items()}
    cfg = O.OracleCfg.from_hf_dict(cfgd, temporal_size=2)
    # (1)
    to = model.clipmodel.text_model(input_ids=ids, attention_mask=mask)
    ref_last, _ = O.text_tower(ids.cpu(), mask.cpu(), sd, cfg)
    assert to.last_hidden_state.shape == ref_last.shape
    assert (to.last_hidden_state.float().cpu() - ref_last).abs().max() < 5e-2 * ref_last.abs().max()
    out = model.clipmodel(input_ids=ids, pixel_values=video, attention_mask=mask)
    want = out.text_embeds @ out.image_embeds.t() * model.clipmodel.logit_scale.exp()
    assert torch.allclose(out.logits_per_text, want, rtol=1e-4, atol=1e-4) and torch.allclose(out["logits_per_image"], want.T, rtol=1e-4, atol=1e-4)
    # the lazily computed logits are differentiable like the reference's (CLIP_ViP.py:1151-1153), through the package's own kernels
    import xpretrain_amd.functional as XF
    te = out.text_embeds.detach().clone().requires_grad_(True)
    ie = out.image_embeds.detach().clone().requires_grad_(True)
    ls = model.clipmodel.logit_scale.detach().clone().requires_grad_(True)
    w = torch.randn(te.shape[0], ie.shape[0], device=te.device)
    (XF.SimLogitsFn.apply(te, ie, ls) * w).sum().backward()
    te2, ie2, ls2 = (t.detach().clone().requires_grad_(True) for t in (te, ie, ls))
    ((te2 @ ie2.t() * ls2.exp()) * w).sum().backward()
    for a, b in ((te, te2), (ie, ie2), (ls, ls2)):
        assert torch.allclose(a.grad, b.grad, rtol=1e-4, atol=1e-4)
    assert out.loss is None
    with pytest.raises(AttributeError):
        out.no_such_field
    # forward-only passes (torch.no_grad) skip the MLP pre-activation output: same features, bit for bit
    with torch.no_grad():
        o_ng = model.clipmodel(input_ids=ids, pixel_values=video, attention_mask=mask)
    assert torch.equal(o_ng.image_embeds, out.image_embeds) and torch.equal(o_ng.text_embeds, out.text_embeds)
    # (2)
    loss_fn = NCELearnableTempLoss()

    def grads():
        for p in model.parameters():
            p.grad = None
        o = model(video, ids, mask)
        loss_fn(o["vis_features"], o["text_features"], model.clipmodel.logit_scale).backward()
        return {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()}
    g_all = grads()
    model.freeze_text_encoder(True)
    g_frozen = grads()
    for n, g in g_frozen.items():
        if ".text_model." in n or "text_projection" in n:
            assert g is None, n
        else:
            assert torch.equal(g, g_all[n]), n
    # (3)
    opt = AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    o = model(video, ids, mask)
    loss = loss_fn(o["vis_features"], o["text_features"], model.clipmodel.logit_scale)
    opt.step()                      # rewrites the bf16 weight copies the graph above saved
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()


def test_fp32_compute_mode_against_oracle():
    """north_star's fp32 leg ("within 1e-3 fp32"): the reference computes fp32 unless fp16 is set (run_pretrain.py:234-236).
    BASELINE config #1 (ViT-B/32, 2 frames 224^2, 16 text tokens, batch 2) in fp32 compute mode -- fp32 MFMA GEMMs,
    the exact-arithmetic attention kernels of csrc/attention_f32.hip, fp32 LayerNorm / embeddings / loss -- against the
    fp32 oracle: features, loss (absolute) and every parameter gradient (relative to the tensor scale)."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import NCELearnableTempLoss
    torch.manual_seed(1234)
    cfgd = O.vit_b_config(patch=32)
    model = VidCLIP(_Args(cfgd, 12))
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
    video, ids, mask = O.synthetic_inputs(2, 2, 224, 16)
    cfg = O.OracleCfg.from_hf_dict(cfgd)
    ref_loss, ref_vis, ref_txt, ref_g = _oracle_step(model, video, ids, mask, cfg, emulate=False)
    model.cuda().train()
    model.clipmodel.set_compute_dtype(torch.float32)
    out = model(video.cuda(), ids.cuda(), mask.cuda())
    loss = NCELearnableTempLoss()(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
    loss.backward()
    dv = (out["vis_features"].cpu() - ref_vis).abs().max().item()
    dt = (out["text_features"].cpu() - ref_txt).abs().max().item()
    dl = abs(loss.item() - ref_loss.item())
    print(f"fp32 mode: loss {loss.item():.6f} oracle {ref_loss.item():.6f} |d loss| {dl:.2e} dvis {dv:.2e} dtxt {dt:.2e}")
    assert dv <= TOL["fp32_abs"] and dt <= TOL["fp32_abs"] and dl <= TOL["fp32_abs"]
    worst = 0.0
    for name, p in model.named_parameters():
        key = name[len("clipmodel."):]
        if ref_g[key].abs().max() <= 1e-6 or name.endswith("k_proj.bias"):
            continue
        worst = max(worst, report(f"fp32 mode grad {name}", p.grad, ref_g[key], 5e-3))
    print(f"fp32 mode: worst gradient error {worst:.2e}")
    assert worst <= 5e-3


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_native_layer_calls_equal_the_op_by_op_path(dtype):
    """xp_encoder_layer_fwd / _bwd (csrc/layer.hip: one C-ABI call per layer pass) issue the same entry points with the same
    arguments as the op-by-op Python path: features, loss and every gradient must be BIT-identical, incl. a frozen subset."""
    import xpretrain_amd.functional as XF
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import NCELearnableTempLoss
    torch.manual_seed(9)
    cfgd = O.vit_b_config(16, 224)
    cfgd["vision_config"]["num_hidden_layers"] = 2
    cfgd["text_config"]["num_hidden_layers"] = 2
    model = VidCLIP(_Args(cfgd, 4)).cuda().train()
    model.clipmodel.set_compute_dtype(dtype)
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
    for n, p in model.named_parameters():          # a frozen subset: the native path skips those gradients by NULL pointers
        if "layers.1.mlp.fc1" in n or "layers.0.layer_norm2" in n or "layers.0.self_attn.k_proj.bias" in n:
            p.requires_grad = False
    video, ids, mask = (t.cuda() for t in O.synthetic_inputs(8 if dtype == torch.bfloat16 else 2, 4, 224, 16))
    loss_fn = NCELearnableTempLoss()

    def run(native):
        old, XF.LAYER_CALLS = XF.LAYER_CALLS, native
        try:
            for p in model.parameters():
                p.grad = None
            out = model(video, ids, mask)
            loss = loss_fn(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
            loss.backward()
            return (out["vis_features"].detach().clone(), out["text_features"].detach().clone(), loss.detach().clone(),
                    {n: (None if p.grad is None else p.grad.clone()) for n, p in model.named_parameters()})
        finally:
            XF.LAYER_CALLS = old
    v1, t1, l1, g1 = run(True)
    v0, t0, l0, g0 = run(False)
    assert torch.equal(v1, v0) and torch.equal(t1, t0) and torch.equal(l1, l0)
    for n in g0:
        assert (g0[n] is None) == (g1[n] is None), n
        if g0[n] is not None:
            assert torch.equal(g0[n], g1[n]), n


def test_gradient_checkpointing_recomputes_bit_identically():
    """CLIPEncoder.gradient_checkpointing (reference CLIP_ViP.py:626,675-690): with it the saved-activation arenas are dropped after the
    forward and every layer is re-run when its backward starts -- same kernels, same inputs: features, loss and all gradients are
    bit-identical to the plain run."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import NCELearnableTempLoss
    torch.manual_seed(7)
    cfgd = O.hf_config_dict(128, 2, 4, 256, 16, 32, 128, 2, 3, 256, 120, 16, 64)
    model = VidCLIP(_Args(cfgd, 3)).cuda().train()
    video, ids, mask = (t.cuda() for t in O.synthetic_inputs(4, 3, 32, 12, vocab=120))

    def run(ckpt):
        if ckpt:
            model.clipmodel.gradient_checkpointing_enable()
        else:
            model.clipmodel.gradient_checkpointing_disable()
        assert all(m.gradient_checkpointing == ckpt for m in model.modules() if hasattr(m, "gradient_checkpointing"))
        for p in model.parameters():
            p.grad = None
        torch.cuda.reset_peak_memory_stats()
        out = model(video, ids, mask)
        loss = NCELearnableTempLoss()(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
        loss.backward()
        torch.cuda.synchronize()
        return (out["vis_features"].detach().clone(), loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()},
                torch.cuda.max_memory_allocated())
    v0, l0, g0, m0 = run(False)
    v1, l1, g1, m1 = run(True)
    assert torch.equal(v0, v1) and torch.equal(l0, l1)
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not bad, bad[:5]
    print(f"peak memory without / with checkpointing: {m0 / 2**20:.0f} / {m1 / 2**20:.0f} MiB (tiny model: cached workspaces dominate)")


@pytest.mark.parametrize("second_stream", ["own", "side"])
@pytest.mark.parametrize("shape", ["tiny", "vit_b_2layers"])
def test_forward_as_two_half_batch_chains_is_bit_identical(monkeypatch, shape, second_stream):
    """functional.ForwardSplit: the video tower's training forward as two half-batch chains on two streams writes the same full-batch
    buffers with the same kernels per row -- features, loss and every gradient are bit-identical to the single chain, repeatedly
    (a race between the chains would show as a run-to-run difference)."""
    import xpretrain_amd.functional as XF
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import NCELearnableTempLoss
    torch.manual_seed(11)
    if shape == "tiny":
        cfgd = O.hf_config_dict(128, 2, 4, 256, 16, 32, 128, 2, 3, 256, 120, 16, 64)
        model = VidCLIP(_Args(cfgd, 3)).cuda().train()
        inputs = O.synthetic_inputs(4, 3, 32, 12, vocab=120)
    else:
        cfgd = O.vit_b_config(16, 224)
        cfgd["vision_config"]["num_hidden_layers"] = 2
        cfgd["text_config"]["num_hidden_layers"] = 1
        model = VidCLIP(_Args(cfgd, 12)).cuda().train()
        inputs = O.synthetic_inputs(4, 12, 224, 32, seed=5)
    video, ids, mask = (t.cuda() for t in inputs)
    monkeypatch.setattr(XF, "FWD_SPLIT_MIN_ROWS", 0)
    monkeypatch.setattr(XF, "FWD_SPLIT_STREAM", second_stream)      # a torch stream of its own / the library's weight-gradient stream
    monkeypatch.setattr(XF, "_SPLIT_STREAMS", {})
    made = []
    real = XF.ForwardSplit

    class Spy(real):
        def __init__(self, device):
            made.append(1)
            super().__init__(device)
    monkeypatch.setattr(XF, "ForwardSplit", Spy)

    def run(on):
        monkeypatch.setattr(XF, "FWD_SPLIT", on)
        for p in model.parameters():
            p.grad = None
        out = model(video, ids, mask)
        loss = NCELearnableTempLoss()(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
        loss.backward()
        torch.cuda.synchronize()
        return out["vis_features"].detach().clone(), loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}
    v0, l0, g0 = run(False)
    assert not made
    for rep in range(3):
        v1, l1, g1 = run(True)
        assert len(made) == rep + 1
        assert torch.equal(v0, v1) and torch.equal(l0, l1)
        bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
        assert not bad, bad[:5]
    with torch.no_grad():               # forward-only passes: two chains as well (the split holds every layer's buffers until the join)
        for rep in range(3):
            f1 = model(video, ids, mask)["vis_features"]
            torch.cuda.synchronize()
            assert len(made) == 4 + rep and torch.equal(f1, v0)
        monkeypatch.setattr(XF, "FWD_SPLIT", False)
        f2 = model(video, ids, mask)["vis_features"]
        assert len(made) == 6 and torch.equal(f2, v0)
    monkeypatch.setattr(XF, "FWD_SPLIT", True)
    for p in model.parameters():        # everything frozen under grad mode: no autograd node keeps the buffers -- the split does
        p.requires_grad_(False)
    f3 = model(video, ids, mask)["vis_features"]
    torch.cuda.synchronize()
    assert len(made) == 7 and torch.equal(f3, v0)


def test_inference_forward_gathers_patches_in_the_gemm(monkeypatch):
    """torch.no_grad() / frozen patch embedding: VisionEmbedFn lets the GEMM loader gather the patches (no im2col pass, no patch
    matrix); the features are bit-identical to the materialised path, for fp32 and uint8 frames; a training pass still materialises
    (dW reads the matrix)."""
    import xpretrain_amd.functional as XF
    from xpretrain_amd import hip_ops as H
    from xpretrain_amd.modeling import VidCLIP
    torch.manual_seed(3)
    cfgd = O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 2, 256, 120, 16, 64)
    model = VidCLIP(_Args(cfgd, 3)).cuda().eval()
    video, ids, mask = (t.cuda() for t in O.synthetic_inputs(2, 3, 32, 12, vocab=120))
    calls = []
    orig = H.im2col
    monkeypatch.setattr(H, "im2col", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    with torch.no_grad():
        a = model.forward_video(video)
        assert not calls                                      # gathered by the loader
        monkeypatch.setattr(XF, "PATCH_GATHER", False)
        b = model.forward_video(video)
        assert calls
        monkeypatch.setattr(XF, "PATCH_GATHER", True)
        u8 = (video.clamp(-1, 1) * 127 + 128).to(torch.uint8)
        c = model.forward_video(u8)
        monkeypatch.setattr(XF, "PATCH_GATHER", False)
        d = model.forward_video(u8)
    assert torch.equal(a, b) and torch.equal(c, d)
    monkeypatch.setattr(XF, "PATCH_GATHER", True)
    calls.clear()
    model.train()
    model(video, ids, mask)["vis_features"].sum().backward()  # training: the matrix is needed for dW
    assert calls and model.clipmodel.vision_model.embeddings.patch_embedding.weight.grad is not None


@pytest.mark.parametrize("first_late", [0, 1, 2])
def test_optimizer_overlapping_the_next_forward_trains_bit_identically(first_late):
    """AdamW.overlap_next_forward: the update of the encoder layers >= K of both towers runs on a stream of its own and the NEXT
    forward waits for it in front of layer K (functional.LATE_WEIGHTS).  Four training steps (4 layers per tower, clip active) must
    leave the plain optimizer's parameters, moments, losses and gradient norms -- the same kernel on the same values; the launch
    partition reorders the norm's chunk partials, so the clip factor may differ in its last bit (1e-6 relative) -- and state_dict()
    readers wait for the late update too (a reader that did not would see the previous step's weights: 1e-3 relative)."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import AdamW, NCELearnableTempLoss, build_e2e_optimizer_w_lr_mul
    from xpretrain_amd import functional as XF
    cfgd = O.hf_config_dict(128, 2, 4, 256, 16, 32, 128, 2, 4, 256, 120, 16, 64)
    video, ids, mask = [t.cuda() for t in O.synthetic_inputs(4, 2, 32, 8, vocab=120)]

    def train(K):
        torch.manual_seed(11)
        model = VidCLIP(_Args(cfgd, 2)).cuda().train()
        groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), 1e-4, 0.05, lr_mul=1, lr_mul_prefix="")
        opt = AdamW([g for g in groups if g["params"]], lr=1e-4, betas=(0.9, 0.98))
        if K is not None:
            opt.overlap_next_forward(model, K)
        loss_fn, trace = NCELearnableTempLoss(), []
        for _ in range(4):          # (a short, stable run: at lr 1e-3 this toy model amplifies the clip factor's last bit to 1e-4 in 4 steps)
            out = model(video, ids, mask)
            loss = loss_fn(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
            loss.backward()
            norm = opt.clip_and_step(0.05)
            if K is not None:
                assert XF.LATE_WEIGHTS["event"] is not None and XF.LATE_WEIGHTS["first_layer"] == K
            opt.zero_grad(set_to_none=True)
            trace.append((loss.detach().clone(), norm.clone()))
        sd = {k: v.clone() for k, v in model.state_dict().items()}                 # (pre-hook: waits for the late update)
        osd = opt.state_dict()["state"]
        moments = [osd[i][k].clone() for i in sorted(osd) for k in ("exp_avg", "exp_avg_sq")]
        XF.LATE_WEIGHTS["event"] = None
        return trace, sd, moments
    t0, sd0, m0 = train(None)
    t1, sd1, m1 = train(first_late)
    for (l0, n0), (l1, n1) in zip(t0, t1):
        torch.testing.assert_close(l1, l0, rtol=2e-6, atol=0)
        torch.testing.assert_close(n1, n0, rtol=2e-6, atol=0)
    for k in sd0:
        if sd0[k].is_floating_point():
            torch.testing.assert_close(sd1[k], sd0[k], rtol=1e-5, atol=1e-7, msg=k)
    assert len(m0) == len(m1)
    for a, b in zip(m0, m1):
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-9)
