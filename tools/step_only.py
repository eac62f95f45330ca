"""Training steps only (no forward probes, no CPU baseline): the workload of bench.py's timed region, for per-kernel A/B profiles
of two builds on one box (run from the root of the tree under test: `python tools/step_only.py [steps]`)."""
import math, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench as B
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from xpretrain_amd import distributed as D
from xpretrain_amd.modeling import VidCLIP
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, get_lr_sched, build_e2e_optimizer_w_lr_mul

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
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
video, ids, mask = O.synthetic_inputs(8, 12, 224, 32, seed=4321)
video, ids, mask = video.to(dev), ids.to(dev), mask.to(dev)
ls = model.clipmodel.logit_scale
n = [1000]


def step():
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    vis, txt = D.gather_features(out["vis_features"], out["text_features"])
    loss = loss_fn(vis, txt, ls)
    loss.backward()
    reducer.synchronize()
    for g in opt.param_groups:
        g["lr"] = get_lr_sched(n[0], "cosine", LR, 100000, warmup_ratio=0.01)
    n[0] += 1
    opt.clip_and_step(5.0)
    reducer.zero_grad()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
print(f"step_only: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step  loss {loss.item():.4f}")
