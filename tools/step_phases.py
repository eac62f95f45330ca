#!/usr/bin/env python
"""Where a data-parallel step spends its time on ONE GPU: bench.py's step with CUDA events on the main stream at the phase boundaries
(forward | loss + backward | reducer.synchronize | clip + AdamW | zero_grad), plain and with the collectives forced through a ONE-rank
RCCL group (feature gather + bucketed gradient all-reduce: no bytes leave the GPU, the streams / events / hooks are the real ones).

    python tools/step_phases.py [steps] [plain|forced|forced_nocomm]
"""
import math
import os
import sys
import time

sys.path.insert(0, os.getcwd())
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
if mode == "forced_nocomm":
    os.environ["XPRETRAIN_DEBUG"] = ",".join(filter(None, [os.environ.get("XPRETRAIN_DEBUG", ""), "no_comm"]))
import torch  # noqa: E402
import bench as B  # noqa: E402
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from xpretrain_amd import distributed as D, functional as XF  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, get_lr_sched, build_e2e_optimizer_w_lr_mul  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if mode != "plain":
    torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:29543", world_size=1, rank=0)
    D.FORCE_COLLECTIVES = True
torch.manual_seed(1234)
model = VidCLIP(B.Args(O.vit_b_config(16, 224)))
with torch.no_grad():
    model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
model.to(dev).train()
D.broadcast_parameters(model)
loss_fn = NCELearnableTempLoss()
reducer = D.GradBucketReducer(model.parameters(), bucket_mb=64.0, average=True, layout_groups=XF.layer_grad_groups(model), segments=D.tower_segments(model))
LR = 5e-6
groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), LR, 0.05, lr_mul=1, lr_mul_prefix="")
opt = AdamW([g for g in groups if g["params"]], lr=LR, betas=(0.9, 0.98))
video, ids, mask = [t.to(dev) for t in O.synthetic_inputs(8, 12, 224, 32, seed=4321)]
ls = model.clipmodel.logit_scale
n = [1000]
NAMES = ["forward", "gather + loss + backward", "reducer.synchronize", "clip + AdamW", "zero_grad"]


def step(ev=None, host=None):
    def mark(i):
        if ev is not None:
            ev[i].record()
            host[i] = time.perf_counter()
    mark(0)
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    mark(1)
    vis, txt = D.gather_features(out["vis_features"], out["text_features"])
    loss_fn(vis, txt, ls).backward()
    mark(2)
    reducer.synchronize()
    mark(3)
    for g in opt.param_groups:
        g["lr"] = get_lr_sched(n[0], "cosine", LR, 100000, warmup_ratio=0.01)
    n[0] += 1
    opt.clip_and_step(5.0)
    mark(4)
    reducer.zero_grad()
    mark(5)


for _ in range(5):
    step()
torch.cuda.synchronize()
gpu = [0.0] * 5
cpu = [0.0] * 5
t0 = time.perf_counter()
s0 = torch.cuda.Event(enable_timing=True); s0.record()
for _ in range(steps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    host = [0.0] * 6
    step(ev, host)
    torch.cuda.synchronize()            # (per step: the phases of one step are then measured from an idle GPU and an idle host)
    for i in range(5):
        gpu[i] += ev[i].elapsed_time(ev[i + 1]) / steps
        cpu[i] += (host[i + 1] - host[i]) * 1e3 / steps
print(f"mode {mode}: main-stream time between phase marks (ms per step; each step starts from an idle GPU) | host time to issue the phase")
for nme, g, c in zip(NAMES, gpu, cpu):
    print(f"  {nme:28s} {g:7.3f} | {c:7.3f}")
print(f"  {'sum':28s} {sum(gpu):7.3f} | {sum(cpu):7.3f}")
# and back to back (the way bench.py measures)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps):
    step()
b.record(); torch.cuda.synchronize()
print(f"  back-to-back: {a.elapsed_time(b) / steps:.3f} ms per step")
if mode != "plain":
    torch.distributed.destroy_process_group()
