#!/usr/bin/env python
"""The ViT-B/16 video tower alone (BASELINE cfg #2 by default), forward only: N passes in inference mode (torch.no_grad)
and N passes in training mode (activations kept for backward), for rocprofv3 --kernel-trace --stats (tools/profile.sh):
the per-kernel table of exactly the passes bench.py reports as `vit_forward_ms` / `vit_forward_train_ms`.
Usage: python tools/fwd_only.py [passes] [frames] [res]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from bench import Args  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 12
res = int(sys.argv[3]) if len(sys.argv) > 3 else 224
mode = sys.argv[4] if len(sys.argv) > 4 else "both"
torch.manual_seed(1234)
model = VidCLIP(Args(O.vit_b_config(16, res))).cuda().train()
video, ids, mask = O.synthetic_inputs(8, frames, res, 32, seed=4321)
video = video.cuda()
vm = model.clipmodel.vision_model


def timed(f):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def infer():
    with torch.no_grad():
        vm(pixel_values=video)


def train():
    vm(pixel_values=video)


if mode in ("both", "infer"):
    print(f"vit forward, inference mode: {timed(infer):.3f} ms")
if mode in ("both", "train"):
    print(f"vit forward, training mode:  {timed(train):.3f} ms")
