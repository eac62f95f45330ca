#!/usr/bin/env python
"""Where the HOST time of a training step goes: cProfile over bench.py's step loop (no device sync inside the loop).
Usage (GPU box): python tools/host_profile.py [steps]"""
import cProfile
import io
import math
import pstats
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

import bench  # noqa: E402
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from xpretrain_amd import distributed as D  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.optimization import AdamW, NCELearnableTempLoss, build_e2e_optimizer_w_lr_mul, get_lr_sched  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = VidCLIP(bench.Args(O.vit_b_config(16, 224))).to(dev).train()
loss_fn = NCELearnableTempLoss()
reducer = D.GradBucketReducer(model.parameters(), bucket_mb=64.0, average=True)
groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), 5e-6, 0.05, lr_mul=1, lr_mul_prefix="")
opt = AdamW([g for g in groups if g["params"]], lr=5e-6, betas=(0.9, 0.98))
video, ids, mask = (t.to(dev) for t in O.synthetic_inputs(8, 12, 224, 32))
ls = model.clipmodel.logit_scale


def step(i):
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    vis, txt = D.gather_features(out["vis_features"], out["text_features"])
    loss = loss_fn(vis, txt, ls)
    loss.backward()
    reducer.synchronize()
    lr = get_lr_sched(1000 + i, "cosine", 5e-6, 100000, warmup_ratio=0.01)
    for g in opt.param_groups:
        g["lr"] = lr
    opt.clip_and_step(5.0)
    reducer.zero_grad()


for i in range(3):
    step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(f"(totals over {steps} steps)")
print("\n".join(line[:150] for line in s.getvalue().splitlines()[:60]))
