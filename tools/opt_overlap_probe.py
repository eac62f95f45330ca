#!/usr/bin/env python
"""What would it buy to run the optimizer update beside the NEXT step's forward?  (round 4 probe; timing only)

The clip + AdamW pass is HBM-bound (5.1 GB, ~0.9 ms) and sits alone at the end of the step; the next forward's first GEMMs are
MFMA-bound.  A pipelined optimizer would update the parameters layer by layer on a second stream while the forward of the next step
runs, each layer's forward waiting only for ITS parameters.  This probe measures the upper bound of that idea without building the
dependency tracking: the optimizer of step k-1 runs on a side stream concurrently with the forward of step k (the forward may read
weights mid-update: numerically meaningless, same kernels and bytes), joined before the backward.

    python tools/opt_overlap_probe.py [steps]
"""
import math
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench as B  # noqa: E402
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from xpretrain_amd import distributed as D  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, build_e2e_optimizer_w_lr_mul  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = VidCLIP(B.Args(O.vit_b_config(16, 224)))
with torch.no_grad():
    model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
model.to(dev).train()
loss_fn = NCELearnableTempLoss()
LR = 5e-6
groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), LR, 0.05, lr_mul=1, lr_mul_prefix="")
opt = AdamW([g for g in groups if g["params"]], lr=LR, betas=(0.9, 0.98))
video, ids, mask = [t.to(dev) for t in O.synthetic_inputs(8, 12, 224, 32, seed=4321)]
ls = model.clipmodel.logit_scale
side = torch.cuda.Stream(device=dev)
held = []


def timed(f, n):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def step_serial():
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    loss_fn(out["vis_features"], out["text_features"], ls).backward()
    opt.clip_and_step(5.0)
    for p in model.parameters():
        p.grad = None


def step_overlapped():
    """the optimizer for the gradients of the PREVIOUS call runs on the side stream beside this call's forward"""
    main = torch.cuda.current_stream()
    if all(p.grad is not None for p in model.parameters() if p.requires_grad):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            opt.clip_and_step(5.0)
        held[:] = [p.grad for p in model.parameters()]      # keep the gradient buffers alive until the join
        for p in model.parameters():
            p.grad = None
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    main.wait_stream(side)
    held.clear()
    loss_fn(out["vis_features"], out["text_features"], ls).backward()


def opt_alone():
    opt.clip_and_step(5.0)


step_serial()
a = timed(step_serial, steps)
# gradients for the first overlapped call
out = model(video, ids, mask)
loss_fn(out["vis_features"], out["text_features"], ls).backward()
b = timed(step_overlapped, steps)
for p in model.parameters():
    if p.grad is None:
        p.grad = torch.zeros_like(p)
c = timed(opt_alone, 5)
print(f"step, optimizer at the end (as bench.py): {a:.3f} ms")
print(f"step, optimizer of the previous step beside the forward (upper bound of a pipelined optimizer): {b:.3f} ms  ({a - b:+.3f} ms)")
print(f"clip + AdamW alone: {c:.3f} ms")
