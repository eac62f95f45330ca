#!/usr/bin/env python
"""Locate run-to-run non-determinism in the full-size (BASELINE cfg #2) step.

Every tensor-producing call of ``xpretrain_amd.hip_ops`` is wrapped; the wrapper keeps a fingerprint of each output of
each call (``--mode sum``: an int64 checksum computed right behind the kernel on its own stream, nothing is kept alive,
the allocator behaves as in production; ``--mode hold``: a clone, so the differing elements can be listed).  The step
is run ``--iters`` times on identical inputs and every run is compared with the first one, call by call in program
order: the FIRST call whose output differs while all earlier outputs agree is the non-deterministic kernel (its inputs
are outputs of earlier calls).  Settings (``--setting``):

    asis       production configuration
    nooverlap  CLIPModel.overlap_text_tower = False  (text tower on the main stream)
    g128       XPRETRAIN_GEMM256=0                   (every GEMM on the 128x128 family)

Usage (GPU box):  python tools/determinism_hunt.py --setting asis --iters 30 --mode sum
"""
import argparse
import collections
import functools
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from xpretrain_amd import workload as O  # noqa: E402  (config + synthetic inputs)
from xpretrain_amd import hip_ops as H  # noqa: E402


class Recorder:
    def __init__(self, mode):
        self.mode = mode
        self.lock = threading.Lock()
        self.reset()

    def reset(self):
        self.items = []          # (label, fingerprint tensor)
        self.counts = collections.Counter()
        self.deferred = []       # (label, tensor) finished by DeferredReduce.flush

    def fp(self, t):
        if self.mode == "hold":
            return t.detach().clone()
        v = t.detach().reshape(-1)
        if v.element_size() == 2:
            return v.view(torch.int16).to(torch.int64).sum()
        if v.element_size() == 4:
            return v.view(torch.int32).to(torch.int64).sum()
        return v.view(torch.int64).sum() if v.element_size() == 8 else v.to(torch.int64).sum()

    def add(self, name, outs, deferred_ptrs):
        with self.lock:
            k = self.counts[name]
            self.counts[name] += 1
        for i, t in enumerate(outs):
            label = f"{name}#{k}.out{i}[{'x'.join(map(str, t.shape))},{str(t.dtype)[6:]}]"
            if t.data_ptr() in deferred_ptrs:
                with self.lock:
                    self.deferred.append((label, t))
            else:
                f = self.fp(t)
                with self.lock:
                    self.items.append((label, f))

    def flush_deferred(self, ptrs):
        with self.lock:
            mine = [(l, t) for l, t in self.deferred if t.data_ptr() in ptrs]
            self.deferred = [(l, t) for l, t in self.deferred if t.data_ptr() not in ptrs]
        for l, t in mine:
            f = self.fp(t)
            with self.lock:
                self.items.append((l + "(deferred)", f))


def tensors_in(x):
    if isinstance(x, torch.Tensor):
        return [x]
    if isinstance(x, (tuple, list)):
        return [t for e in x for t in tensors_in(e)]
    return []


def install(rec):
    skip = {"workspace", "cast_back", "gemm_auto_split", "colsum_deferred"}
    names = [n for n, f in vars(H).items() if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == H.__name__
             and not isinstance(f, type) and n not in skip]
    for n in names:
        orig = getattr(H, n)

        def make(orig, n):
            @functools.wraps(orig)
            def wrapper(*a, **kw):
                tag = n
                if n == "gemm":
                    tag = f"gemm[M{a[2]},N{a[3]},K{a[4]},aks{int(kw.get('a_kstrided', False))},bks{int(kw.get('b_kstrided', False))},epi{kw.get('epilogue', 0)},sk{kw.get('split_k', 1)}]"
                if rec.mode == "hold":      # the inputs as they are when the call is made (corruption between production and use)
                    ins = {"layernorm_bwd": lambda: list(a[:5]) + [kw.get("dres")],
                           "gemm": lambda: [a[0], a[1], kw.get("resid"), kw.get("bias")],
                           "attn_bwd": lambda: list(a[:4]),
                           "layernorm_fwd": lambda: list(a[:3])}.get(n, lambda: [])()
                    rec.add(tag + ".IN", [t for t in ins if isinstance(t, torch.Tensor)], set())
                r = orig(*a, **kw)
                outs = [t for t in tensors_in(r) if t.dtype != torch.uint8]      # uint8 = untyped workspaces
                seen = {t.data_ptr() for t in outs}
                for key in ("out", "dx", "dgamma", "dbeta", "aux"):
                    t = kw.get(key)
                    if isinstance(t, torch.Tensor) and t.data_ptr() not in seen and t.dtype != torch.uint8:
                        outs.append(t)
                if n == "gemm" and kw.get("epilogue", 0) == 6:     # patch embed: proxy rows of `out` are written by vip_proxy_rows
                    outs = []
                dptr = set()
                for key in ("defer", "colsum_defer"):
                    d = kw.get(key)
                    if d is not None:
                        dptr |= {o.data_ptr() for _, o in d._keep}
                rec.add(tag, outs, dptr)
                return r
            return wrapper
        setattr(H, n, make(orig, n))
    # colsum_deferred: its output is final only after flush
    orig_cd = H.colsum_deferred

    def cd(X, rows, cols, defer, ldx=None, name="colsum"):
        out = orig_cd(X, rows, cols, defer, ldx=ldx, name=name)
        rec.add("colsum_deferred", [out], {out.data_ptr()})
        return out
    H.colsum_deferred = cd
    orig_flush = H.DeferredReduce.flush

    def flush(self):
        ptrs = {o.data_ptr() for _, o in self._keep}
        orig_flush(self)
        rec.flush_deferred(ptrs)
    H.DeferredReduce.flush = flush


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--setting", default="asis", choices=["asis", "nooverlap", "g128"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--mode", default="sum", choices=["sum", "hold", "none"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "determinism"))
    args = ap.parse_args()
    if args.setting == "g128":
        os.environ["XPRETRAIN_GEMM256"] = "0"

    from tests.test_model_gpu import _Args
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.modeling.CLIP_ViP import CLIPModel
    from xpretrain_amd.optimization import NCELearnableTempLoss
    if args.setting == "nooverlap":
        CLIPModel.overlap_text_tower = False

    torch.manual_seed(1234)
    model = VidCLIP(_Args(O.vit_b_config(16, 224), 12)).cuda().train()
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.02)
    video, ids, mask = (t.cuda() for t in O.synthetic_inputs(args.batch, 12, 224, 32))
    loss_fn = NCELearnableTempLoss()
    rec = Recorder(args.mode)
    if args.mode != "none":
        install(rec)

    def step():
        rec.reset()
        for p in model.parameters():
            p.grad = None
        out = model(video, ids, mask)
        loss = loss_fn(out["vis_features"], out["text_features"], model.clipmodel.logit_scale)
        loss.backward()
        torch.cuda.synchronize()
        items = list(rec.items)
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        return items, grads

    step()                                        # warm-up (weight casts, workspace growth)
    ref_items, ref_grads = step()
    ref = {l: f for l, f in ref_items}
    order = [l for l, _ in ref_items]
    summary = {"setting": args.setting, "mode": args.mode, "iters": args.iters, "calls_per_step": len(order),
               "bad_iters": 0, "first_diff": collections.Counter(), "grad_diff_iters": 0, "detail": []}
    for it in range(args.iters):
        items, grads = step()
        cur = {l: f for l, f in items}
        gbad = [n for n in ref_grads if not torch.equal(ref_grads[n], grads[n]) and not n.endswith("token_embedding.weight")]
        # text tower (side stream) and vision tower calls interleave differently run to run: compare by label
        bad = [l for l in order if l in cur and not torch.equal(ref[l], cur[l])]
        missing = [l for l in order if l not in cur]
        if gbad:
            summary["grad_diff_iters"] += 1
        if bad or missing:
            summary["bad_iters"] += 1
            first = bad[0] if bad else "missing:" + missing[0]
            summary["first_diff"][first.split("#")[0]] += 1
            d = {"iter": it, "first": first, "n_bad_calls": len(bad), "bad_head": bad[:6], "grads_bad": gbad[:4], "n_grads_bad": len(gbad)}
            if args.mode == "hold" and bad:
                d["bad_inputs"] = [l for l in bad if ".IN#" in l][:6]
                a, b = ref[first], cur[first]
                ne = (a != b)
                idx = ne.nonzero()
                d["n_diff_elems"] = int(ne.sum())
                d["diff_idx_head"] = idx[:12].tolist()
                if a.is_floating_point():
                    d["max_abs_diff"] = float((a.float() - b.float()).abs().max())
                    d["a_vals"] = a[ne][:6].float().tolist()
                    d["b_vals"] = b[ne][:6].float().tolist()
                if idx.numel() and idx.shape[1] == 2:
                    d["rows_span"] = [int(idx[:, 0].min()), int(idx[:, 0].max())]
                    d["cols_span"] = [int(idx[:, 1].min()), int(idx[:, 1].max())]
            summary["detail"].append(d)
            print("DIFF", json.dumps(d), flush=True)
        elif gbad:
            print("GRAD-ONLY DIFF", it, gbad[:4], flush=True)
    summary["first_diff"] = dict(summary["first_diff"])
    os.makedirs(args.out, exist_ok=True)
    fn = os.path.join(args.out, f"hunt_{args.setting}_{args.mode}.json")
    with open(fn, "w") as f:
        json.dump(summary, f, indent=1)
    print(f"[{args.setting}/{args.mode}] {summary['bad_iters']}/{args.iters} iterations differ "
          f"({summary['grad_diff_iters']} with differing parameter gradients); first-diff histogram: {summary['first_diff']}")


if __name__ == "__main__":
    main()
