#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ from the UNMODIFIED reference.

Run in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

The reference (microsoft/XPretrain CLIP-ViP) has no tests or golden vectors of its own
(SURVEY.md §4), so these fixtures -- outputs of the reference itself on seeded inputs --
are what pins ``oracle/clipvip_oracle.py`` and, through it, the HIP path.  The GPU box
has no /root/reference; it only ever reads the committed ``*.pt`` files.

Fixtures (all fp32, CPU, torch.manual_seed'ed):
  tiny_e2e.pt        tiny CLIP-ViP (2 vision layers/128 wide/2 heads, 2 text layers) through
                     reference VidCLIP.forward + NCELearnableTempLoss + backward: state dict,
                     inputs, features, hidden states, loss, every parameter gradient.
  attn_forward2.pt   CLIPAttention.forward2 alone for several (M,N,L), with input grads.
  text_attn.pt       CLIPAttention.forward (causal + ragged padding masks), with input grads.
  temporal_interp.pt CLIPVisionViPEmbeddings for T in {1,8,12,32} (temporal_size 12).
  loss.pt            NCELearnableTempLoss / NCELearnableTempLoss_vsc_fc for several batch sizes
                     and logit scales (incl. both clamp limits 0 and ln 200).
  retrieval.pt       src/utils/metrics.py (cal_cossim, np_softmax, compute_metrics, compute_metrics_multi) in the simple
                     and DSL settings of validate(), incl. exact score ties; ImageNorm arithmetic on uint8 frames.
  full_cfg2.pt, full_cfg3.pt, full_cfg4.pt, full_cfg2_b8.pt (configs[1] at the bench batch of 8: features, loss, gradients)
                     BASELINE configs[1], [3], [4] at FULL model size (ViT-B/16; 12x224^2 / 8x448^2 / 32x224^2, 32 text
                     tokens) and batch 2 through the reference VidCLIP.forward + NCELearnableTempLoss + backward, weights
                     rebuilt from seeds (tests/gpu_util.py::seeded_model): features, loss, sampled rows of every hidden
                     state, every 1-D gradient, three rows of every weight gradient.  (Batch 2: at batch 1 the contrastive
                     loss is identically 0.)
  optim.pt           src/optimization: AdamW.step x 6 with clip_grad_norm_ 5.0 and the warmup-cosine schedule
                     over the four build_e2e_optimizer_w_lr_mul groups; get_lr_sched tables.
"""
import math
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from oracle import clipvip_oracle as O  # noqa: E402

TINY = dict(vision_hidden=128, vision_heads=2, vision_layers=2, vision_inter=192, patch=8, image=32,
            text_hidden=128, text_heads=2, text_layers=2, text_inter=192, vocab=120, max_pos=16, proj=64)


def randomize_(model, seed):
    """Reference init leaves biases 0, LN affine at (1,0) and temporal_embedding 0
    (CLIP_ViP.py:166,481-522) -- perturb them so every term is exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("logit_scale"):
                continue
            if n.endswith(".bias") or "temporal_embedding" in n:
                p.add_(0.02 * torch.randn(p.shape, generator=g))
            elif "layer_norm" in n or "layrnorm" in n or "layernorm" in n:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() >= 2 and "embedding" not in n:
                # reference init stds are ~0.01-0.06 at this width; widen so attention is not ~uniform
                p.mul_(3.0)


def tiny_e2e(ref):
    torch.manual_seed(1234)
    cfg_dict = O.hf_config_dict(**TINY)
    args = ref_import.make_args(cfg_dict, add_cls_num=3, temporal_size=3)
    model = ref.VidCLIP.VidCLIP(args)
    randomize_(model, 99)
    model.train()
    B, T, Lt = 4, 3, 12
    video, ids, mask = O.synthetic_inputs(B, T, TINY["image"], Lt, vocab=TINY["vocab"], seed=4321)
    # one row with EOT in the very last slot and one with the earliest legal EOT
    ids[0, 2:] = TINY["vocab"] - 1; mask[0] = 0; mask[0, :3] = 1
    ids[1, 1:-1] = torch.randint(1, TINY["vocab"] - 2, (Lt - 2,)); ids[1, -1] = TINY["vocab"] - 1; mask[1] = 1
    out = model(video, ids, mask)
    loss_fn = ref.loss.NCELearnableTempLoss(None)
    loss = loss_fn(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
    loss.backward()
    # hidden states through the reference vision/text towers
    with torch.no_grad():
        vo = model.clipmodel.vision_model(pixel_values=video, output_hidden_states=True, return_dict=True)
        to = model.clipmodel.text_model(input_ids=ids, attention_mask=mask, output_hidden_states=True,
                                        return_dict=True)
    fx = dict(
        config=cfg_dict, add_cls_num=3, temporal_size=3,
        state_dict={k: v.detach().clone() for k, v in model.state_dict().items()},
        video=video, ids=ids, mask=mask,
        vis_features=out["vis_features"].detach(), text_features=out["text_features"].detach(),
        loss=loss.detach(),
        grads={n: p.grad.detach().clone() for n, p in model.named_parameters()},
        vision_hidden=[h.detach() for h in vo.hidden_states], vision_last=vo.last_hidden_state.detach(),
        vision_pooled=vo.pooler_output.detach(),
        text_hidden=[h.detach() for h in to.hidden_states], text_last=to.last_hidden_state.detach(),
        text_pooled=to.pooler_output.detach(),
    )
    torch.save(fx, os.path.join(HERE, "tiny_e2e.pt"))
    print("tiny_e2e: loss", float(loss), "params", sum(p.numel() for p in model.parameters()))


def full_size(ref, name, frames, res, B=2, txt_len=32, patch=16, temporal_size=12, hidden=True):
    """One full-size case; what is kept is small (see the module docstring).  Uses this repo's model class only to BUILD
    the seeded weights -- the numbers stored come from the reference."""
    from tests.gpu_util import seeded_model, sample_rows
    cfgd = O.vit_b_config(patch, res)
    ours = seeded_model(cfgd, temporal_size)
    args = ref_import.make_args(cfgd, add_cls_num=3, temporal_size=temporal_size)
    model = ref.VidCLIP.VidCLIP(args)
    model.load_state_dict(ours.state_dict(), strict=True)
    del ours
    model.train()
    video, ids, mask = O.synthetic_inputs(B, frames, res, txt_len)
    out = model(video, ids, mask)
    loss = ref.loss.NCELearnableTempLoss(None)(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
    loss.backward()
    with torch.no_grad():
        vo = model.clipmodel.vision_model(pixel_values=video, output_hidden_states=True, return_dict=True)
        to = model.clipmodel.text_model(input_ids=ids, attention_mask=mask, output_hidden_states=True, return_dict=True)
        # calibration: the reference's OWN reduced-precision path on the same input -- torch.autocast(bfloat16) (pure
        # .bfloat16() fails in its text tower, SURVEY 8a defect 2).  How far bf16 matmul inputs alone move features, logits
        # and loss away from the fp32 run is what "2e-2 bf16" can mean at logit scale e^4.6 ~ 100.
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ob = model(video, ids, mask)
        bv, bt = ob["vis_features"].float(), ob["text_features"].float()
        sc = model.clipmodel.logit_scale.exp()
        ref_bf16 = dict(loss=ref.loss.NCELearnableTempLoss(None)(bv, bt, model.clipmodel.logit_scale).item(),
                        dvis=(bv - out["vis_features"]).abs().max().item(), dtxt=(bt - out["text_features"]).abs().max().item(),
                        dlogits=((bv @ bt.t() - out["vis_features"] @ out["text_features"].t()) * sc).abs().max().item())
    L = (res // patch) ** 2
    S = 4 + frames * L
    rows = sample_rows(S, 4)
    grads = {}
    for n, p in model.named_parameters():
        g = p.grad.detach()
        if g.dim() <= 1 or g.numel() <= 4096 * 4:
            grads[n] = g.clone()
        elif n.endswith("token_embedding.weight"):
            grads[n + "#rows"] = (ids.unique(), g[ids.unique()].clone())
        else:
            g2 = g.reshape(g.shape[0], -1)
            pick = torch.tensor([0, g2.shape[0] // 2, g2.shape[0] - 1])
            grads[n + "#rows"] = (pick, g2[pick].clone())
    fx = dict(patch=patch, frames=frames, res=res, B=B, txt_len=txt_len, temporal_size=temporal_size, rows=rows,
              vis_features=out["vis_features"].detach(), text_features=out["text_features"].detach(), loss=loss.detach(),
              vision_hidden=[h[:, rows].detach().half() for h in vo.hidden_states] if hidden else None,
              text_hidden=[h.detach().half() for h in to.hidden_states] if hidden else None,
              vision_pooled=vo.pooler_output.detach(), text_pooled=to.pooler_output.detach(), grads=grads, ref_bf16=ref_bf16)
    torch.save(fx, os.path.join(HERE, name))
    print(name, "reference bf16 autocast vs fp32:", ref_bf16)
    print(name, "loss", float(loss), "size MB", os.path.getsize(os.path.join(HERE, name)) / 1e6)


def _attn_module(ref, D, H, seed):
    from transformers.models.clip.configuration_clip import CLIPVisionConfig
    torch.manual_seed(seed)
    cfg = CLIPVisionConfig(hidden_size=D, num_attention_heads=H, intermediate_size=2 * D, num_hidden_layers=1)
    m = ref.CLIP_ViP.CLIPAttention(cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * (0.5 / math.sqrt(D) if p.dim() == 2 else 0.1))
        m.q_proj.weight.mul_(4.0)   # sharpen the softmax
    return m


def attn_forward2(ref):
    cases = []
    for (M, N, L), D, H, B in [((4, 2, 49), 128, 2, 2), ((4, 2, 196), 64, 1, 1), ((1, 3, 5), 64, 1, 2),
                               ((4, 3, 70), 128, 2, 1), ((2, 5, 16), 192, 3, 2)]:
        m = _attn_module(ref, D, H, 7 + M + N + L)
        S = M + N * L
        x = torch.randn(B, S, D, requires_grad=True)
        y = m.forward2(x, (M, N, L))
        gy = torch.randn_like(y)
        y.backward(gy)
        cases.append(dict(size=(M, N, L), D=D, H=H, x=x.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(),
                          sd={k: v.detach().clone() for k, v in m.state_dict().items()}))
    torch.save(cases, os.path.join(HERE, "attn_forward2.pt"))
    print("attn_forward2:", [c["size"] for c in cases])


def text_attn(ref):
    cases = []
    enc = ref.CLIP_ViP.CLIPTextTransformer
    for B, S, D, H, mode in [(3, 12, 128, 2, "ragged"), (2, 32, 64, 1, "ragged"), (2, 7, 64, 1, "none"),
                             (2, 16, 64, 1, "allpad")]:
        m = _attn_module(ref, D, H, 31 + S)
        x = torch.randn(B, S, D, requires_grad=True)
        causal = enc._build_causal_attention_mask(None, B, S)
        if mode == "none":
            mask, am = None, None
        else:
            lens = torch.randint(1, S + 1, (B,))
            lens[0] = S
            mask = (torch.arange(S)[None] < lens[:, None]).long()
            if mode == "allpad":
                mask[1] = 0          # degenerate row: every key padded (finfo.min on all of them)
            am = ref.CLIP_ViP._expand_mask(mask, x.dtype)
        y, _ = m(x, None, attention_mask=am, causal_attention_mask=causal)
        gy = torch.randn_like(y)
        y.backward(gy)
        cases.append(dict(D=D, H=H, mask=mask, x=x.detach(), y=y.detach(), gy=gy, gx=x.grad.clone(),
                          sd={k: v.detach().clone() for k, v in m.state_dict().items()}))
    torch.save(cases, os.path.join(HERE, "text_attn.pt"))
    print("text_attn:", len(cases))


def temporal_interp(ref):
    from transformers.models.clip.configuration_clip import CLIPVisionConfig
    torch.manual_seed(5)
    cfg = CLIPVisionConfig(hidden_size=64, num_attention_heads=1, intermediate_size=128, num_hidden_layers=1,
                           image_size=16, patch_size=8)
    add = ref.AttrDict(type="ViP", temporal_size=12, if_use_temporal_embed=1, logit_scale_init_value=4.6,
                       add_cls_num=3)
    emb = ref.CLIP_ViP.CLIPVisionViPEmbeddings(cfg, add)
    with torch.no_grad():
        emb.temporal_embedding.copy_(torch.randn_like(emb.temporal_embedding))
        emb.patch_embedding.weight.mul_(0.05)
    cases = []
    for T in (1, 8, 12, 32):
        emb.zero_grad()
        v = torch.randn(2, T, 3, 16, 16)
        y, size = emb(v)
        gy = torch.randn_like(y)
        y.backward(gy)
        cases.append(dict(T=T, video=v, y=y.detach(), size=tuple(size), gy=gy,
                          g_temporal=emb.temporal_embedding.grad.clone(),
                          g_patch=emb.patch_embedding.weight.grad.clone(),
                          g_pos=emb.position_embedding.weight.grad.clone(),
                          g_cls=emb.class_embedding.grad.clone(), g_added=emb.added_cls.grad.clone()))
    torch.save(dict(sd={k: v.detach().clone() for k, v in emb.state_dict().items()}, cases=cases),
               os.path.join(HERE, "temporal_interp.pt"))
    print("temporal_interp:", [c["T"] for c in cases])


def loss(ref):
    torch.manual_seed(11)
    cases = []
    f1 = ref.loss.NCELearnableTempLoss(None)
    f2 = ref.loss.NCELearnableTempLoss_vsc_fc(None)
    for n, d in [(2, 64), (8, 512), (16, 128), (64, 64), (5, 32)]:
        for ls in (0.0, 4.6, math.log(200.0)):
            feats = [torch.nn.functional.normalize(torch.randn(n, d), dim=-1).requires_grad_() for _ in range(4)]
            t = torch.tensor(ls, requires_grad=True)
            l1 = f1(feats[0], feats[1], t)
            g1 = torch.autograd.grad(l1, [feats[0], feats[1], t])
            l2 = f2(feats[0], feats[1], feats[2], feats[3], t)
            g2 = torch.autograd.grad(l2, feats + [t])
            cases.append(dict(n=n, d=d, log_scale=ls, feats=[f.detach() for f in feats],
                              nce=l1.detach(), nce_grads=[g.clone() for g in g1],
                              vsc_fc=l2.detach(), vsc_fc_grads=[g.clone() for g in g2]))
    torch.save(cases, os.path.join(HERE, "loss.pt"))
    print("loss:", len(cases))


def optim(ref):
    import importlib
    import warnings
    adamw = importlib.import_module("src.optimization.adamw")
    sched = importlib.import_module("src.optimization.sched")
    utils = importlib.import_module("src.optimization.utils")
    torch.manual_seed(77)
    names = ["clipmodel.vision_model.encoder.layers.0.mlp.fc1.weight", "clipmodel.vision_model.encoder.layers.0.mlp.fc1.bias",
             "clipmodel.text_model.encoder.layers.0.self_attn.q_proj.weight", "clipmodel.text_model.final_layer_norm.bias",
             "clipmodel.logit_scale", "clipmodel.visual_projection.weight", "clipmodel.vision_model.embeddings.added_cls"]
    shapes = [(48, 33), (48,), (64, 64), (64,), (), (32, 48), (3, 37)]
    params = [torch.nn.Parameter(torch.randn(s) * 0.05) for s in shapes]
    init = [p.detach().clone() for p in params]
    lr0, wd, lr_mul, steps, total = 1e-3, 0.2, 0.1, 6, 20
    groups = utils.build_e2e_optimizer_w_lr_mul(list(zip(names, params)), lr0, wd, lr_mul=lr_mul,
                                                lr_mul_prefix="text_model")
    group_idx = [[[id(p) for p in params].index(id(q)) for q in g["params"]] for g in groups]
    opt = adamw.AdamW(groups, lr=lr0, betas=(0.9, 0.98))
    grads, norms, lrs = [], [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(steps):
            lr_t = sched.get_lr_sched(t, "cosine", lr0, total, warmup_ratio=0.2)
            for gi, g in enumerate(opt.param_groups):                 # run_video_retrieval.py:377-383
                g["lr"] = lr_mul * lr_t if gi in (0, 1) else lr_t
            gs = [torch.randn(s) * (4.0 if t % 2 == 0 else 0.03) for s in shapes]  # clip active on even steps only
            for p, g in zip(params, gs):
                p.grad = g.clone()
            norms.append(torch.nn.utils.clip_grad_norm_(params, 5.0).clone())
            opt.step()
            grads.append(gs)
            lrs.append(lr_t)
    table = {d: [sched.get_lr_sched(t, d, 3e-4, 50, warmup_ratio=0.1) for t in range(0, 56)]
             for d in ("linear", "cosine", "invsqrt", "constant")}
    torch.save(dict(names=names, init=init, grads=grads, norms=norms, lrs=lrs, lr_mul=lr_mul, weight_decay=wd,
                    betas=(0.9, 0.98), group_idx=group_idx, final=[p.detach().clone() for p in params],
                    exp_avg=[opt.state[p]["exp_avg"].clone() for p in params],
                    exp_avg_sq=[opt.state[p]["exp_avg_sq"].clone() for p in params], sched_table=table),
               os.path.join(HERE, "optim.pt"))
    print("optim: steps", steps, "norms", [round(float(n), 3) for n in norms])


def retrieval(ref):
    """src/utils/metrics.py on seeded feature sets: plain, with exact duplicate captions (ties), multi-label."""
    import importlib
    import numpy as np
    metrics = importlib.import_module("src.utils.metrics")
    rng = np.random.RandomState(5)
    cases = []
    for n, d, noise, dup in [(64, 32, 1.5, 0), (200, 64, 2.2, 0), (333, 48, 2.5, 0), (96, 16, 0.8, 7)]:
        vis = rng.randn(n, d).astype(np.float32)
        txt = vis + noise * rng.randn(n, d).astype(np.float32)
        if dup:                                   # duplicated videos: exact score ties in every text row
            vis[n - dup:] = vis[:dup]
        vis /= np.linalg.norm(vis, axis=1, keepdims=True)
        txt /= np.linalg.norm(txt, axis=1, keepdims=True)
        sim = metrics.cal_cossim(txt, vis)
        res = {}
        s2 = sim
        for setting in ("simple", "DSL"):
            if setting == "DSL":
                s2 = s2 * metrics.np_softmax(s2 * 100, axis=0)        # run_video_retrieval.py:170-171
            res[setting] = dict(v2t=metrics.compute_metrics(s2.T), t2v=metrics.compute_metrics(s2))
        labels = rng.randint(0, n, size=n)
        cases.append(dict(txt=torch.from_numpy(txt), vis=torch.from_numpy(vis),
                          sim=torch.from_numpy(sim) if n <= 96 else None,
                          softmax100=torch.from_numpy(metrics.np_softmax(sim * 100, axis=0)) if n <= 96 else None, results=res,
                          labels=torch.from_numpy(labels), multi=metrics.compute_metrics_multi(sim, labels.tolist())))
    # ImageNorm on the device (data_utils.py:256-281) is CUDA-only; its arithmetic is three in-place tensor ops, run here
    # on CPU tensors with the same operation order: div_(255.), sub_(mean), div_(std)
    g = torch.Generator().manual_seed(8)
    frames = torch.randint(0, 256, (2, 3, 3, 32, 32), generator=g, dtype=torch.uint8)
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
    img = frames.float()
    img.div_(255.)
    img = img.sub_(torch.tensor(mean).view(1, 1, 3, 1, 1)).div_(torch.tensor(std).view(1, 1, 3, 1, 1))
    torch.save(dict(cases=cases, frames=frames, normed=img, mean=mean, std=std), os.path.join(HERE, "retrieval.pt"))
    print("retrieval:", [(c["results"]["simple"]["t2v"][0], c["results"]["DSL"]["t2v"][0]) for c in cases])


if __name__ == "__main__":
    ref = ref_import.load()
    if len(sys.argv) > 1 and sys.argv[1] in ("optim", "retrieval"):
        {"optim": optim, "retrieval": retrieval}[sys.argv[1]](ref)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "full_b8":     # the bench batch (8 pairs = 256 text rows): features, loss, gradients
        full_size(ref, "full_cfg2_b8.pt", 12, 224, B=8, hidden=False)
        sys.exit(0)
    elif len(sys.argv) > 1 and sys.argv[1] == "full":        # minutes of CPU time each; not part of the default regeneration
        for nm, fr, rs in (("full_cfg2.pt", 12, 224), ("full_cfg3.pt", 8, 448), ("full_cfg4.pt", 32, 224)):
            if len(sys.argv) < 3 or sys.argv[2] in nm:
                full_size(ref, nm, fr, rs)
        sys.exit(0)
    tiny_e2e(ref)
    attn_forward2(ref)
    text_attn(ref)
    temporal_interp(ref)
    loss(ref)
    optim(ref)
    retrieval(ref)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".pt"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
