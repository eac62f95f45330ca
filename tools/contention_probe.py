#!/usr/bin/env python
"""One-GPU proxy for the gradient all-reduce's cost beside the backward pass: bench.py's training step alone, and
with a memory-bound copy kernel confined to C workgroups (xp_probe_stream_copy -- a ring collective's channel kernels stream the
gradient buckets the same way) running on a second stream from the start of backward until the optimizer step.
Usage: python tools/contention_probe.py [steps]"""
import ctypes as C
import math
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from bench import Args  # noqa: E402
from xpretrain_amd import distributed as D, _lib as L  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, build_e2e_optimizer_w_lr_mul  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["thin", "fat"]                  # which aggressors
counts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 16, 32, 64]    # workgroups per aggressor
torch.manual_seed(1234)
dev = torch.device("cuda", 0)
model = VidCLIP(Args(O.vit_b_config(16, 224))).to(dev).train()
loss_fn = NCELearnableTempLoss()
reducer = D.GradBucketReducer(model.parameters(), bucket_mb=64.0, average=True)
groups = build_e2e_optimizer_w_lr_mul(list(model.named_parameters()), 5e-6, 0.05, lr_mul=1, lr_mul_prefix="")
opt = AdamW([g for g in groups if g["params"]], lr=5e-6, betas=(0.9, 0.98))
video, ids, mask = [t.to(dev) for t in O.synthetic_inputs(8, 12, 224, 32, seed=4321)]
ls = model.clipmodel.logit_scale
side = torch.cuda.Stream(device=dev)
nbytes = 598 << 20                                       # the fp32 gradient volume of the model
src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
dst = torch.empty_like(src)


def step(blocks=0, iters=1, fn="xp_probe_stream_copy"):
    with torch.no_grad():
        ls.clamp_(0, math.log(200.0))
    out = model(video, ids, mask)
    loss = loss_fn(out["vis_features"], out["text_features"], ls)
    if blocks:
        side.wait_stream(torch.cuda.current_stream())
        L.check(getattr(L.lib(), fn)(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, blocks, iters,
                                     C.c_void_p(side.cuda_stream)), fn)
    loss.backward()
    if blocks:
        torch.cuda.current_stream().wait_stream(side)
    opt.clip_and_step(5.0)
    reducer.zero_grad()


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


base = timed(step, steps)
print(f"step alone: {base:.3f} ms")
# how many streams can the step touch before the runtime's stream -> hardware-queue mapping bites?  (main, second forward chain, text
# tower, weight-gradient stream are in use already; a collective adds at least one.)  Extra streams that do NOTHING but an event
# hand-shake with the main stream at the start of backward:
import os  # noqa: E402
for n_extra in [int(x) for x in os.environ.get("PROBE_EXTRA_STREAMS", "").split(",") if x]:
    extra = [torch.cuda.Stream(device=dev) for _ in range(n_extra)]

    def step_touch():
        with torch.no_grad():
            ls.clamp_(0, math.log(200.0))
        out = model(video, ids, mask)
        loss = loss_fn(out["vis_features"], out["text_features"], ls)
        main = torch.cuda.current_stream()
        for st in extra:
            st.wait_stream(main)
        loss.backward()
        for st in extra:
            main.wait_stream(st)
        opt.clip_and_step(5.0)
        reducer.zero_grad()
    t = timed(step_touch, steps)
    print(f"step with {n_extra} extra idle stream(s) forked at the start of backward and joined after it: {t:.3f} ms ({t - base:+.3f} ms)")
# "thin": a copy kernel with a handful of registers (fits beside a GEMM workgroup's waves); "rccl footprint": the same copy in a
# kernel with the resources of RCCL's gfx950 collective kernel (csrc/probe.hip: 288 registers per lane = one wave per SIMD, 19,744 B
# LDS) -- it cannot share a CU with a 256x256-GEMM workgroup, so its workgroups take whole CUs away from the GEMMs for as long as
# the copy lasts, which is what a bucket all-reduce beside the backward pass does.
print(f"CU budget of the split-K planning: {L.lib().xp_get_cu_budget()} (XPRETRAIN_CU_BUDGET / distributed.reserve_cus_for_collectives)")
for fn, label in (("xp_probe_stream_copy", "thin"), ("xp_probe_stream_copy_fat", "rccl footprint")):
    if ("thin" if label == "thin" else "fat") not in kinds:
        continue
    for blocks in counts:
        def alone():
            L.check(getattr(L.lib(), fn)(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), nbytes, blocks, 1,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "copy")
        t_copy = timed(alone, 5)
        # 2 x 7/8 x 598 MB is what a ring all-reduce moves per GPU on 8 GPUs: one read + one write pass of the buffer ~ iters = 1
        t = timed(lambda: step(blocks, 1, fn), steps)
        print(f"{label:14s} {blocks:3d} copy workgroups: copy alone {t_copy:.3f} ms ({2 * nbytes / t_copy / 1e6:.0f} GB/s read+write); step with "
              f"the copy beside backward {t:.3f} ms (+{t - base:.3f} ms, {100 * (t - base) / base:.1f} %); serial sum would be {base + t_copy:.3f} ms")
