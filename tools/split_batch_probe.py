#!/usr/bin/env python
"""Does running the local batch as two half-batches on two HIP streams pay?  (round 4 experiment)

The forward of the video tower is a strict chain per sample; its memory-bound kernels (LayerNorm, attention, GEMM epilogues)
and its MFMA-bound main loops never overlap on one stream.  Two independent half-batch chains on two streams, skewed against
each other, let one chain's memory-bound kernels run beside the other's GEMMs.  This probe measures, on one box:
  * the ViT forward in training mode: full batch on one stream vs two half-batches on two streams (several skews);
  * the whole training step: the bench.py step vs the same step with the model applied to two half-batches on two streams and the
    features concatenated before the loss (autograd then accumulates the two weight gradients with an add kernel per parameter --
    an overhead a native implementation would not have, so the step number is a lower bound of the gain).

    python tools/split_batch_probe.py [--steps 10]
"""
import argparse
import math
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench as B  # noqa: E402
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from xpretrain_amd import distributed as D  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, get_lr_sched, build_e2e_optimizer_w_lr_mul  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = VidCLIP(B.Args(O.vit_b_config(16, 224)))
with torch.no_grad():
    model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
model.to(dev).train()
loss_fn = NCELearnableTempLoss()
reducer = D.GradBucketReducer(model.parameters(), bucket_mb=64.0, average=True)
LR = 5e-6
groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), LR, 0.05, lr_mul=1, lr_mul_prefix="")
opt = AdamW([g for g in groups if g["params"]], lr=LR, betas=(0.9, 0.98))
video, ids, mask = [t.to(dev) for t in O.synthetic_inputs(8, 12, 224, 32, seed=4321)]
ls = model.clipmodel.logit_scale
sA, sB = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
vm = model.clipmodel.vision_model
n = [1000]


def timed(f, iters):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def fwd_full():
    return vm(pixel_values=video)


def fwd_split(skew_us=0):
    main = torch.cuda.current_stream()
    sA.wait_stream(main); sB.wait_stream(main)
    with torch.cuda.stream(sA):
        a = vm(pixel_values=video[:4])
    with torch.cuda.stream(sB):
        if skew_us:
            torch.cuda._sleep(int(skew_us * 2100))          # ~cycles at 2.1 GHz
        b = vm(pixel_values=video[4:])
    main.wait_stream(sA); main.wait_stream(sB)
    return a, b


print(f"ViT forward, training mode, full batch on one stream: {timed(fwd_full, 10):.3f} ms")
for skew in (0, 100, 200, 400):
    print(f"ViT forward, two half-batches on two streams, second one delayed by ~{skew} us: {timed(lambda: fwd_split(skew), 10):.3f} ms")
with torch.no_grad():
    print(f"ViT forward, inference mode, full batch: {timed(fwd_full, 10):.3f} ms")
    for skew in (0, 200):
        print(f"ViT forward, inference mode, two half-batches, skew ~{skew} us: {timed(lambda: fwd_split(skew), 10):.3f} ms")


def finish(vis, txt):
    loss = loss_fn(vis, txt, ls)
    loss.backward()
    reducer.synchronize()
    for g in opt.param_groups:
        g["lr"] = get_lr_sched(n[0], "cosine", LR, 100000, warmup_ratio=0.01)
    n[0] += 1
    opt.clip_and_step(5.0)
    reducer.zero_grad()


def step_full():
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    finish(out["vis_features"], out["text_features"])


def step_split(skew_us=200):
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    main = torch.cuda.current_stream()
    sA.wait_stream(main); sB.wait_stream(main)
    with torch.cuda.stream(sA):
        oa = model(video[:4], ids[:4], mask[:4])
    with torch.cuda.stream(sB):
        if skew_us:
            torch.cuda._sleep(int(skew_us * 2100))
        ob = model(video[4:], ids[4:], mask[4:])
    main.wait_stream(sA); main.wait_stream(sB)
    vis = torch.cat([oa["vis_features"], ob["vis_features"]])
    txt = torch.cat([oa["text_features"], ob["text_features"]])
    finish(vis, txt)
    main.wait_stream(sA); main.wait_stream(sB)


print(f"training step, full batch: {timed(step_full, args.steps):.3f} ms")
for skew in (0, 200):
    print(f"training step, two half-batches on two streams (+ one gradient-accumulate kernel per parameter), skew ~{skew} us: "
          f"{timed(lambda: step_split(skew), args.steps):.3f} ms")
print(f"training step, full batch (again): {timed(step_full, args.steps):.3f} ms")
