"""GPU: the RCCL code path of distributed.py on ONE GPU -- a 1-rank `nccl` process group with the collectives forced on:
packed feature all-gather (+ its collective-free backward and the verify switch), bucketed asynchronous gradient
all-reduce driven by autograd hooks, parameter broadcast, then clip + AdamW reading the flat bucket views.  The result
must equal the plain single-process step.  (Multi-rank semantics are covered on gloo by tests/test_distributed_cpu.py.)"""
import os

import pytest
import torch

from oracle import clipvip_oracle as O

pytestmark = pytest.mark.gpu


def _step(model, opt, reducer, D, batch):
    from xpretrain_amd.optimization import NCELearnableTempLoss
    video, ids, mask = batch
    out = model(video, ids, mask)
    vis, txt = D.gather_features(out["vis_features"], out["text_features"], verify_identical=True)
    loss = NCELearnableTempLoss()(vis, txt, model.clipmodel.logit_scale)
    loss.backward()
    reducer.synchronize()
    norm = opt.clip_and_step(1.0)
    reducer.zero_grad()
    return loss.detach().clone(), norm.detach().clone()


def test_one_rank_nccl_group_matches_plain_step():
    import torch.distributed as dist
    from xpretrain_amd import distributed as D
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.optimization import AdamW
    from tests.test_model_gpu import _Args
    cfgd = O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 2, 256, 120, 16, 64)
    video, ids, mask = O.synthetic_inputs(4, 2, 32, 8, vocab=120)
    batch = (video.cuda(), ids.cuda(), mask.cuda())

    def run(distributed, sinks=False, wire=None):
        import xpretrain_amd.functional as XF
        torch.manual_seed(5)
        model = VidCLIP(_Args(cfgd, 2)).cuda().train()
        D.broadcast_parameters(model)
        reducer = D.GradBucketReducer(model.parameters(), bucket_mb=0.25, average=True,     # several buckets
                                      layout_groups=XF.layer_grad_groups(model) if sinks else None, wire_dtype=wire,
                                      segments=D.tower_segments(model) if sinks else None)
        assert reducer._active == distributed
        if sinks:
            assert len(XF.GRAD_SINKS) == 4            # 2 video + 2 text layers publish a gradient sink each
        opt = AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
        res = []
        for it in range(3):
            if sinks and it == 1:
                # direct writes: after backward every layer parameter's .grad already IS its bucket view (no pack copy)
                video, ids, mask = batch
                out = model(video, ids, mask)
                vis, txt = D.gather_features(out["vis_features"], out["text_features"])
                from xpretrain_amd.optimization import NCELearnableTempLoss
                NCELearnableTempLoss()(vis, txt, model.clipmodel.logit_scale).backward()
                for b in reducer.buckets:
                    for p, v in zip(b["params"], b["views"]):
                        if any(p is q for g in XF.layer_grad_groups(model) for q in g):
                            assert p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                reducer.synchronize(); reducer.zero_grad()
            res.append(_step(model, opt, reducer, D, batch))
        reducer.remove()
        assert not XF.GRAD_SINKS
        return res, {k: v.detach().clone() for k, v in model.state_dict().items()}

    def dual_pass_grads(sinks):
        """VidCLIP.forward's second (image, caption) pass applies every encoder layer TWICE in one graph (VidCLIP.py:70-79): with
        gradient sinks on, only the first application of a layer may write the sink (functional._claim_sink) -- the gradients must
        be g1 + g2 exactly as without sinks, not two aliases of the second write."""
        import xpretrain_amd.functional as XF
        from xpretrain_amd.optimization import NCELearnableTempLoss_vsc_fc
        torch.manual_seed(5)
        model = VidCLIP(_Args(cfgd, 2)).cuda().train()
        reducer = D.GradBucketReducer(model.parameters(), bucket_mb=0.25, average=True,
                                      layout_groups=XF.layer_grad_groups(model) if sinks else None,
                                      segments=D.tower_segments(model) if sinks else None)
        video, ids, mask = batch
        image = video[:, :1].contiguous()
        cap_ids, cap_mask = ids.flip(0).contiguous().view(-1, 1, ids.shape[1]), mask.flip(0).contiguous().view(-1, 1, mask.shape[1])
        out = model(video, ids, mask, image=image, caption_ids=cap_ids, caption_masks=cap_mask)
        loss = NCELearnableTempLoss_vsc_fc()(out["vis_features"], out["text_features"], out["img_features"], out["cap_features"],
                                             model.clipmodel.logit_scale)
        loss.backward()
        reducer.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        reducer.zero_grad(); reducer.remove()
        return loss.detach().clone(), grads

    plain, sd0 = run(False)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", world_size=1, rank=0)
    try:
        D.FORCE_COLLECTIVES = True
        forced, sd1 = run(True)
        direct, sd2 = run(True, sinks=True)           # gradients written straight into the bucket storage
        wire16, sd3 = run(True, sinks=True, wire=torch.bfloat16)
        dl0, dg0 = dual_pass_grads(False)
        dl1, dg1 = dual_pass_grads(True)
    finally:
        D.FORCE_COLLECTIVES = False
        dist.destroy_process_group()
    for (l0, n0), (l1, n1) in zip(plain, forced):
        assert abs(l0.item() - l1.item()) <= 1e-5 * max(1.0, abs(l0.item()))
        assert abs(n0.item() - n1.item()) <= 1e-4 * max(1.0, abs(n0.item()))
    for k in sd0:
        assert torch.allclose(sd0[k].float(), sd1[k].float(), rtol=1e-5, atol=1e-6), k
    # the sink run did one extra backward at it == 1 without stepping: its three optimizer steps must equal the forced run's
    for (l1, n1), (l2, n2) in zip(forced, direct):
        assert l1.item() == l2.item() and n1.item() == n2.item()
    for k in sd1:
        assert torch.equal(sd1[k], sd2[k]), k
    # bf16 on the wire: gradients rounded once to bf16 (2^-9 relative) before the reduction; three AdamW steps at lr 1e-3 on a
    # random tiny model amplify that (measured 3e-3 of the loss after the third step)
    for it, ((l1, n1), (l3, n3)) in enumerate(zip(forced, wire16)):
        assert abs(l1.item() - l3.item()) <= (1e-5 if it == 0 else 1e-2) * max(1.0, abs(l1.item()))
        assert abs(n1.item() - n3.item()) <= 5e-2 * abs(n1.item())
    # the dual pass with sinks: every gradient equals the no-sink run (a layer applied twice writes its sink once)
    assert dl0.item() == dl1.item() and dg0.keys() == dg1.keys()
    bad = [n for n in dg0 if not torch.allclose(dg0[n], dg1[n], rtol=1e-6, atol=1e-9)]
    assert not bad, bad[:5]
