#!/usr/bin/env python
"""In-step calibration of the 256-wide GEMM kernels against the vendor library (VERDICT r3, next #1a).  TOOLS ONLY: nothing in the
product path calls the vendor library.

The training step of bench.py runs op by op (XPRETRAIN_DEBUG=op_by_op, so that every xp_gemm call of the encoder layers passes
through hip_ops.gemm); every video-tower GEMM (token dimension >= 4096) gets a SHADOW: the same product computed by torch.mm
(hipBLASLt) on the same operands into a scratch buffer, launched right before or right after our kernel on the same stream.  Both
launches are bracketed by HIP events on that stream.  The step is longer than the real one (twice the GEMM work), but each vendor
kernel runs in the neighbourhood our kernel runs in -- same operands in the same cache / HBM state, same clock and power
regime -- which is the comparison isolated micro-benchmarks cannot give.

Our kernels do more than the vendor's in several calls (fused bias / quick_gelu with two outputs / residual add / gelu' / fp32
split-K slabs + a separate reduce); the table lists the epilogue so the rows can be read for what they are.

    python tools/vendor_instep.py [--steps 6] [--order after|before|both] [--out gpurun_out/vendor_instep.txt]
"""
import argparse
import math
import os
import statistics
import sys

os.environ["XPRETRAIN_DEBUG"] = ",".join(filter(None, [os.environ.get("XPRETRAIN_DEBUG", ""), "op_by_op"]))
sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench as B  # noqa: E402
from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from xpretrain_amd import distributed as D, hip_ops as H, _lib as L  # noqa: E402
from xpretrain_amd.modeling import VidCLIP  # noqa: E402
from xpretrain_amd.optimization import NCELearnableTempLoss, AdamW, get_lr_sched, build_e2e_optimizer_w_lr_mul  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--order", default="both")
ap.add_argument("--out", default=None)
ap.add_argument("--no-shadow", action="store_true", help="events only (how much the bracketing itself costs)")
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
n = [1000]

EPI = {L.EPI_NONE: "-", L.EPI_BIAS: "bias", L.EPI_BIAS_QSCALE: "bias+qscale", L.EPI_BIAS_GELU: "bias+gelu, 2 outputs",
       L.EPI_BIAS_RESID: "bias+residual", L.EPI_GELU_BWD: "x gelu'(pre) (+colsum)"}
_gemm = H.gemm
_reduce = H.splitk_reduce
records = {}          # key -> {"own": [ms...], "vendor": [ms...]}
pending = []          # (key, which, start event, end event)
scratch = {}
state = {"order": "after", "on": False}


def _scratch(shape, dtype):
    t = scratch.get((shape, dtype))
    if t is None:
        t = scratch[(shape, dtype)] = torch.empty(shape, dtype=dtype, device=dev)
    return t


def _timed(key, which, fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    pending.append((key, which, s, e))
    return r


def gemm(A, Bm, M, N, K, **kw):
    big = max(M, K) >= 4096 and min(M, N, K) >= 512 and A.dtype == torch.bfloat16 and not kw.get("a_remap", (0, 0, 0))[0]
    if not state["on"] or not big or torch.cuda.current_stream() != torch.cuda.default_stream():
        return _gemm(A, Bm, M, N, K, **kw)
    aks, bks = kw.get("a_kstrided", False), kw.get("b_kstrided", False)
    kind = "dW" if aks else ("dX" if bks else "fwd")
    split = kw.get("split_k", 1)
    key = (kind, M, N, K, EPI.get(kw.get("epilogue", L.EPI_NONE), "?") + (f", split-K {split} fp32 slabs" if split > 1 else ""))

    def vendor():
        if args.no_shadow:
            return
        a = A.view(K, M).t() if aks else A.view(M, K)
        b = Bm.view(K, N) if bks else Bm.view(N, K).t()
        torch.mm(a, b, out=_scratch((M, N), torch.bfloat16))
    if state["order"] == "before":
        _timed(key, "vendor", vendor)
    out = _timed(key, "own", lambda: _gemm(A, Bm, M, N, K, **kw))
    if state["order"] == "after":
        _timed(key, "vendor", vendor)
    return out


def splitk_reduce(slabs, out, accumulate=False, splits=None):
    if not state["on"] or torch.cuda.current_stream() != torch.cuda.default_stream() or out.numel() < 512 * 512:
        return _reduce(slabs, out, accumulate, splits)
    return _timed(("dW-reduce", out.shape[0], out.shape[1], 0, f"splitk_reduce x{splits}"), "own",
                  lambda: _reduce(slabs, out, accumulate, splits))


H.gemm = gemm
H.splitk_reduce = splitk_reduce


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


def run(order):
    state["order"], state["on"] = order, True
    records.clear()
    for it in range(2 + args.steps):
        pending.clear()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        step()
        t1.record()
        torch.cuda.synchronize()
        if it < 2:
            continue
        for key, which, s, e in pending:
            records.setdefault(key, {"own": [], "vendor": []})[which].append(s.elapsed_time(e) * 1e3)
    state["on"] = False
    lines = [f"order: vendor launched {order} our kernel; {args.steps} steps x 12 layers per row; HIP-event brackets on the launch stream (us, median); "
             f"instrumented step {t0.elapsed_time(t1):.2f} ms"]
    lines.append(f"{'kind':9s} {'M':>6s} {'N':>6s} {'K':>6s}  {'ours us':>8s} {'TF/s':>6s}  {'vendor us':>9s} {'TF/s':>6s}  vendor/ours  our epilogue")
    tot_o = tot_v = 0.0
    for key in sorted(records, key=lambda k: ("fwd", "dX", "dW", "dW-reduce").index(k[0]) * 10 ** 9 + k[1] * k[2]):
        kind, M, N, K, epi = key
        o = statistics.median(records[key]["own"])
        if kind == "dW-reduce":
            lines.append(f"{kind:9s} {M:6d} {N:6d} {'':>6s}  {o:8.1f} {'':>6s}  {'':>9s} {'':>6s}  {'':>11s}  {epi}")
            continue
        v = statistics.median(records[key]["vendor"]) if records[key]["vendor"] else float("nan")
        fl = 2.0 * M * N * K
        lines.append(f"{kind:9s} {M:6d} {N:6d} {K:6d}  {o:8.1f} {fl / o / 1e6:6.0f}  {v:9.1f} {fl / v / 1e6:6.0f}  {v / o:11.3f}  {epi}")
        tot_o += o; tot_v += v
    lines.append(f"sum over the 12 GEMMs of a layer: ours {tot_o:.1f} us (+ the dW reduces above), vendor {tot_v:.1f} us (plain products only)")
    return "\n".join(lines)


for _ in range(3):
    step()
torch.cuda.synchronize()
report = []
for order in (("after", "before") if args.order == "both" else (args.order,)):
    report.append(run(order))
text = "\n\n".join(report)
print(text)
if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        f.write(text + "\n")
