#!/usr/bin/env python
"""What does fc1's second output cost?    The training forward's fc1 GEMM writes `pre` (the backward's gelu'
argument) beside `act`: 231 of its 265 MB.  (1) the GEMM alone, with and without the aux output, sustained clocks; (2) the video tower's
training-mode forward (two half-batch chains, as in the step) with XPRETRAIN_DEBUG=fc1_no_pre -- the layer call drops the aux pointer,
every other kernel unchanged -- against the normal forward, interleaved rounds on one box.  Forward only: without `pre` the backward
would read garbage.   python tools/fc1_one_output.py [rounds] [passes]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L, workload as O  # noqa: E402
from bench import Args  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M, D, Dff = 8 * 2356, 768, 3072
bf = torch.bfloat16
A = torch.randn(M, D, device="cuda").to(bf); W = (torch.randn(Dff, D, device="cuda") * 0.02).to(bf)
bias = torch.zeros(Dff, device="cuda"); out = torch.empty(M, Dff, dtype=bf, device="cuda"); aux = torch.empty_like(out)


def timeit(fn, iters, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


for r in range(3):
    two = timeit(lambda: H.gemm(A, W, M, Dff, D, out=out, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux), 200, 300) * 1e3
    one = timeit(lambda: H.gemm(A, W, M, Dff, D, out=out, epilogue=L.EPI_BIAS_GELU, bias=bias), 200, 300) * 1e3
    print(f"fc1 GEMM alone, round {r}: two outputs {two:6.1f} us | one output {one:6.1f} us | difference {two - one:5.1f} us")

torch.manual_seed(1234)
model = VidCLIP(Args(O.vit_b_config(16, 224))).cuda().train()
video = O.synthetic_inputs(8, 12, 224, 32, seed=4321)[0].cuda()
vm = model.clipmodel.vision_model
res = {"": [], "fc1_no_pre": []}
for r in range(rounds):
    for flag in ("", "fc1_no_pre"):
        os.environ["XPRETRAIN_DEBUG"] = flag
        res[flag].append(timeit(lambda: vm(pixel_values=video), passes, 10 if r else 40))
os.environ["XPRETRAIN_DEBUG"] = ""
print("ViT training-mode forward, ms per pass, interleaved rounds:")
print("  two outputs (pre + act):", " ".join(f"{t:.3f}" for t in res[""]), f"| median {sorted(res[''])[rounds // 2]:.3f}")
print("  one output  (act only): ", " ".join(f"{t:.3f}" for t in res["fc1_no_pre"]), f"| median {sorted(res['fc1_no_pre'])[rounds // 2]:.3f}")
