#!/usr/bin/env python
"""Which Python call sites launch torch fill / zero / copy kernels inside one training step?  (65 FillFunctor
launches per step.)  Runs bench.py's step under torch.profiler with stacks and prints the aten ops by call site."""
import collections
import math
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from bench import Args  # noqa: E402
from xpretrain_amd import distributed as D  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, build_e2e_optimizer_w_lr_mul  # noqa: E402

torch.manual_seed(1234)
dev = torch.device("cuda", 0)
model = VidCLIP(Args(O.vit_b_config(16, 224))).to(dev).train()
loss_fn = NCELearnableTempLoss()
reducer = D.GradBucketReducer(model.parameters(), bucket_mb=64.0, average=True)
groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), 5e-6, 0.05, lr_mul=1, lr_mul_prefix="")
opt = AdamW([g for g in groups if g["params"]], lr=5e-6, betas=(0.9, 0.98))
video, ids, mask = [t.to(dev) for t in O.synthetic_inputs(8, 12, 224, 32, seed=4321)]
ls = model.clipmodel.logit_scale


def step():
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    vis, txt = D.gather_features(out["vis_features"], out["text_features"])
    loss = loss_fn(vis, txt, ls)
    loss.backward()
    reducer.synchronize()
    opt.clip_and_step(5.0)
    reducer.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::ones_like", "aten::mul", "aten::add_",
                  "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::cat", "aten::stack"):
        st = [f for f in (e.stack or []) if "xpretrain_amd" in f or "bench" in f or "find_fills" in f or "autograd" in f]
        sites[(e.name, tuple(st[:3]))] += 1
for (name, st), n in sites.most_common(40):
    print(f"{n:4d} x {name:18s} {' <- '.join(s.strip()[-90:] for s in st) or '(no python frame: autograd engine / C++)'}")
print("--- GPU kernels of the step that are not ours:")
k = collections.Counter(e.name[:100] for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "at::native" in e.name)
for name, n in k.most_common(15):
    print(f"{n:4d} x {name}")
