#!/usr/bin/env python
"""Where the streams of the UNPROFILED step are, back to back (no per-step synchronisation, no profiler): events recorded on the main
stream and on the text tower's stream at the host-side phase boundaries of bench.py's step, plus events at every encoder layer of the
video tower's backward (a tensor hook on each layer's input gradient, recorded by the autograd thread on the stream it runs that node on).
Read out after the last step; one line per step with every mark as an offset (ms) from the step's first mark.

    python tools/step_events.py [steps]

A profiler trace of this step is host-bound (21 ms under rocprofv3 against 14.5 ms) and shows the text tower's backward running
alone after the video tower's; this shows where it runs when the host is ahead of the GPU, as it is in the timed region."""
import math
import os
import statistics
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench as B  # noqa: E402
from xpretrain_amd import workload as O  # noqa: E402
from xpretrain_amd import distributed as D, functional as XF  # noqa: E402,F401
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.modeling.CLIP_ViP import CLIPModel  # noqa: E402
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, get_lr_sched, build_e2e_optimizer_w_lr_mul  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
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
ovl = [a for a in sys.argv if a.startswith("--opt-overlap")]
if ovl:         # --opt-overlap[=K]: AdamW.overlap_next_forward(model, K)
    opt.overlap_next_forward(model, int(ovl[0].partition("=")[2] or "3"))
video, ids, mask = [t.to(dev) for t in O.synthetic_inputs(8, 12, 224, 32, seed=4321)]
ls = model.clipmodel.logit_scale
n = [1000]
text_stream = CLIPModel.shared_text_stream(dev)
marks = []          # per step: list of (name, event)


def mark(name, stream=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else torch.cuda.current_stream())
    marks[-1].append((name, e))


# the video tower's backward, layer by layer: a hook on the gradient of each layer's INPUT fires on the autograd thread right after
# the layer's backward node was issued; the event lands on the stream that node ran on
vis_layers = model.clipmodel.vision_model.encoder.layers
txt_layers = model.clipmodel.text_model.encoder.layers


def layer_hooks(layers, tag):
    for i, layer in enumerate(layers):
        def pre(mod, args, kwargs=None, i=i):
            x = args[0] if args else None
            if torch.is_tensor(x) and x.requires_grad:
                x.register_hook(lambda g, i=i: (mark(f"{tag}bwd{i}"), None)[1])
        layer.register_forward_pre_hook(pre)


def step():
    marks.append([])
    mark("start")
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    mark("fwd")
    mark("fwd_text", text_stream)
    vis, txt = D.gather_features(out["vis_features"], out["text_features"])
    loss = loss_fn(vis, txt, ls)
    mark("loss")
    loss.backward()
    mark("bwd_main")
    mark("bwd_text", text_stream)
    reducer.synchronize()
    for g in opt.param_groups:
        g["lr"] = get_lr_sched(n[0], "cosine", LR, 100000, warmup_ratio=0.01)
    n[0] += 1
    opt.clip_and_step(5.0)
    mark("opt")
    if opt._late_stream is not None:
        mark("late_end", opt._late_stream)        # the overlapped update of THIS step's gradients: ends inside the next step
    reducer.zero_grad()
    return loss


for _ in range(4):
    step()
hooked = "--layers" in sys.argv
if hooked:
    layer_hooks(vis_layers, "v")
    layer_hooks(txt_layers, "t")
torch.cuda.synchronize()
marks.clear()
for _ in range(steps):
    step()
torch.cuda.synchronize()
names = [nme for nme, _ in marks[-1]]
print("offsets (ms) from each step's first mark; back-to-back steps, no profiler" + (" (layer hooks on: the split forward's layer order)" if hooked else ""))
rows = []
for k, m in enumerate(marks[2:-1]):
    base = m[0][1]
    d = {nme: base.elapsed_time(e) for nme, e in m}
    nxt = marks[2 + k + 1][0][1]
    d["next_start"] = base.elapsed_time(nxt)
    rows.append(d)
keys = [k for k in rows[0]]
for k in keys:
    v = [r[k] for r in rows if k in r]
    print(f"  {k:12s} median {statistics.median(v):8.3f}   min {min(v):8.3f}  max {max(v):8.3f}")
