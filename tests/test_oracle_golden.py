"""CPU: pin oracle/clipvip_oracle.py against the fixtures produced by the UNMODIFIED
reference (tests/golden/make_golden.py), and against the live reference when present."""
import math

import pytest
import torch

from oracle import clipvip_oracle as O
from oracle import ref_import

TOL = dict(rtol=1e-4, atol=5e-5)   # fp32 roundoff between two op orders; values are O(1-10)


def assert_maxrel(a, b, tol, name="", atol=1e-5):
    """max|a-b| <= tol * max(|b|, tiny): error relative to the tensor's scale (fp32 roundoff of two
    different summation orders does not respect elementwise rtol near zero crossings).  atol covers gradients that are
    mathematically zero (k_proj.bias: softmax is shift-invariant) and so are pure roundoff."""
    err = (a - b).abs().max().item()
    scale = max(b.abs().max().item(), 1e-12)
    assert err <= tol * scale + atol, f"{name}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def _cfg(fx):
    return O.OracleCfg.from_hf_dict(fx["config"], add_cls_num=fx["add_cls_num"], temporal_size=fx["temporal_size"])


def test_tiny_e2e_forward_hidden_states(golden):
    fx = golden("tiny_e2e.pt")
    cfg = _cfg(fx)
    sd = O.strip_prefix(fx["state_dict"])
    vh, th = [], []
    vlast, vpool = O.vision_tower(fx["video"], sd, cfg, collect=vh)
    # reference hidden_states = [post pre_layrnorm, after layer 1, ..]; ours collects [embed, preLN, layers..]
    for ours, ref in zip(vh[1:], fx["vision_hidden"]):
        torch.testing.assert_close(ours, ref, **TOL)
    torch.testing.assert_close(vlast, fx["vision_last"], **TOL)
    torch.testing.assert_close(vpool, fx["vision_pooled"], **TOL)
    tlast, tpool = O.text_tower(fx["ids"], fx["mask"], sd, cfg, collect=th)
    for ours, ref in zip(th, fx["text_hidden"]):
        torch.testing.assert_close(ours, ref, **TOL)
    torch.testing.assert_close(tlast, fx["text_last"], **TOL)
    torch.testing.assert_close(tpool, fx["text_pooled"], **TOL)


def test_tiny_e2e_loss_and_all_grads(golden):
    fx = golden("tiny_e2e.pt")
    cfg = _cfg(fx)
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in O.strip_prefix(fx["state_dict"]).items()}
    loss, vis, txt = O.full_step(fx["video"], fx["ids"], fx["mask"], sd, cfg)
    torch.testing.assert_close(vis, fx["vis_features"], **TOL)
    torch.testing.assert_close(txt, fx["text_features"], **TOL)
    torch.testing.assert_close(loss, fx["loss"], rtol=1e-5, atol=1e-5)
    loss.backward()
    assert len(fx["grads"]) > 50
    for name, g in fx["grads"].items():
        ours = sd[name[len("clipmodel."):]].grad
        assert ours is not None, name
        assert_maxrel(ours, g, 2e-4, name)


def test_forward2_cases(golden):
    for c in golden("attn_forward2.pt"):
        x = c["x"].clone().requires_grad_()
        y = O.attention_block(x, c["sd"], "", c["H"], c["size"], None)
        torch.testing.assert_close(y, c["y"], **TOL)
        y.backward(c["gy"])
        torch.testing.assert_close(x.grad, c["gx"], rtol=1e-4, atol=1e-5)


def test_forward2_equals_block_masked_dense():
    torch.manual_seed(0)
    for (M, N, L) in [(4, 2, 49), (4, 3, 10), (1, 4, 7)]:
        S = M + N * L
        q, k, v = (torch.randn(2, 3, S, 16, dtype=torch.float64) for _ in range(3))
        a = O.proxy_attention_core(q, k, v, (M, N, L))
        b = O.proxy_attention_core_masked(q, k, v, (M, N, L))
        torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-12)


def test_text_attention_cases(golden):
    for c in golden("text_attn.pt"):
        x = c["x"].clone().requires_grad_()
        y = O.attention_block(x, c["sd"], "", c["H"], None, c["mask"])
        torch.testing.assert_close(y, c["y"], **TOL)
        y.backward(c["gy"])
        torch.testing.assert_close(x.grad, c["gx"], rtol=1e-4, atol=1e-5)


def test_temporal_interpolation(golden):
    fx = golden("temporal_interp.pt")
    sd = {"vision_model.embeddings." + k: v for k, v in fx["sd"].items()}
    cfg = O.OracleCfg(O.TowerCfg(64, 1, 1, 128), O.TowerCfg(64, 1, 1, 128), patch=8, image=16, add_cls_num=3,
                      temporal_size=12, proj=64)
    for c in fx["cases"]:
        sdg = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        y, size = O.vip_embeddings(c["video"], sdg, cfg)
        assert tuple(size) == c["size"]
        torch.testing.assert_close(y, c["y"], **TOL)
        y.backward(c["gy"])
        p = "vision_model.embeddings."
        torch.testing.assert_close(sdg[p + "temporal_embedding"].grad, c["g_temporal"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sdg[p + "patch_embedding.weight"].grad, c["g_patch"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(sdg[p + "position_embedding.weight"].grad, c["g_pos"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sdg[p + "class_embedding"].grad, c["g_cls"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(sdg[p + "added_cls"].grad, c["g_added"], rtol=1e-4, atol=1e-5)


def test_losses(golden):
    for c in golden("loss.pt"):
        feats = [f.clone().requires_grad_() for f in c["feats"]]
        t = torch.tensor(c["log_scale"], requires_grad=True)
        l1 = O.nce_learnable_temp_loss(feats[0], feats[1], t)
        torch.testing.assert_close(l1, c["nce"], rtol=1e-5, atol=1e-5)
        for a, b in zip(torch.autograd.grad(l1, [feats[0], feats[1], t]), c["nce_grads"]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        l2 = O.nce_vsc_fc_loss(*feats, t)
        torch.testing.assert_close(l2, c["vsc_fc"], rtol=1e-5, atol=1e-5)
        for a, b in zip(torch.autograd.grad(l2, feats + [t]), c["vsc_fc_grads"]):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


@pytest.mark.skipif(not ref_import.available(), reason="reference tree not present (GPU box)")
def test_live_reference_cfg1_shape():
    """Live cross-check at BASELINE cfg #1's architecture (ViT-B/32, T=2, Lt=16) with B=2."""
    ref = ref_import.load()
    torch.manual_seed(1234)
    cfgd = O.vit_b_config(patch=32)
    model = ref.VidCLIP.VidCLIP(ref_import.make_args(cfgd, add_cls_num=3, temporal_size=12))
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
    video, ids, mask = O.synthetic_inputs(2, 2, 224, 16)
    out = model(video, ids, mask)
    loss = ref.loss.NCELearnableTempLoss(None)(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
    cfg = O.OracleCfg.from_hf_dict(cfgd)
    sd = O.strip_prefix(model.state_dict())
    with torch.no_grad():
        l2, vis, txt = O.full_step(video, ids, mask, sd, cfg)
    torch.testing.assert_close(vis, out["vis_features"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(txt, out["text_features"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(l2, loss, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("fixture", ["full_cfg2.pt", "full_cfg2_b8.pt"])
def test_oracle_matches_reference_at_full_size_cfg2(golden, fixture):
    """The restatement against the unmodified reference at BASELINE configs[1]'s full model size (ViT-B/16, 12 frames 224^2,
    32 text tokens; tests/golden/full_cfg2.pt at batch 2, full_cfg2_b8.pt at the bench batch of 8): features, loss, sampled
    hidden-state rows (batch 2) and gradients, fp32."""
    from tests.gpu_util import seeded_model
    fx = golden(fixture)
    cfgd = O.vit_b_config(fx["patch"], fx["res"])
    model = seeded_model(cfgd, fx["temporal_size"])
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in O.strip_prefix(model.state_dict()).items()}
    del model
    video, ids, mask = O.synthetic_inputs(fx["B"], fx["frames"], fx["res"], fx["txt_len"])
    cfg = O.OracleCfg.from_hf_dict(cfgd, temporal_size=fx["temporal_size"])
    vh = []
    _, vp = O.vision_tower(video, sd, cfg, collect=vh)
    _, tp = O.text_tower(ids, mask, sd, cfg)
    vis = O.l2_normalize(vp @ sd["visual_projection.weight"].t())
    txt = O.l2_normalize(tp @ sd["text_projection.weight"].t())
    loss = O.nce_learnable_temp_loss(vis, txt, sd["logit_scale"])
    loss.backward()
    assert (vis - fx["vis_features"]).abs().max() < 2e-5 and (txt - fx["text_features"]).abs().max() < 2e-5
    assert abs(loss.item() - fx["loss"].item()) < 1e-4
    for h, r in zip(vh[1:], fx["vision_hidden"] or []):    # fixture rows are stored in fp16 (the batch-8 fixture keeps none)
        assert (h[:, fx["rows"]].detach() - r.float()).abs().max() <= 2e-3 * max(1.0, r.float().abs().max().item())
    for key, ref in fx["grads"].items():
        if key.endswith("#rows"):
            pick, ref = ref
            g = sd[key[len("clipmodel."):-5]].grad
            g = g.reshape(g.shape[0], -1)[pick]
        else:
            g = sd[key[len("clipmodel."):]].grad
        scale = ref.abs().max().item()
        if scale < 1e-7:          # k_proj.bias: mathematically zero (softmax is shift-invariant), pure rounding noise
            continue
        assert (g - ref).abs().max().item() <= 2e-3 * scale, key
