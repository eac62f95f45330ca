#!/usr/bin/env python
"""Kernel-level bitwise harness: a VICTIM kernel at BASELINE cfg #2 shapes is run ``--iters`` times on identical inputs
on the main stream (outputs and workspaces NaN-poisoned before every launch) while an AGGRESSOR kernel loops on a side
stream; every victim output is compared bit for bit with the first run.  Localises (a) races / uninitialised reads
inside one kernel (aggressor ``none``) and (b) interference between concurrently running kernels.

    python tools/race_repro.py --victims ln_bwd,gemm_dx --aggressors none,ln_text,gemm_text,attn_text --iters 2000
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from xpretrain_amd import hip_ops as H, _lib as L  # noqa: E402

BF = torch.bfloat16
ROWS, D, DFF, B, S, HEADS = 18848, 768, 3072, 8, 2356, 12
SIZE = (4, 12, 196)


def poison(*ts):
    for t in ts:
        if t.is_floating_point():
            t.fill_(float("nan"))
        else:
            t.fill_(-1)


def make_victims(dev):
    g = torch.Generator(device=dev).manual_seed(7)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
    x = rn(ROWS, D).to(BF)
    dy = rn(ROWS, D, sc=1e-3).to(BF)
    dres = rn(ROWS, D, sc=1e-3).to(BF)
    gamma = 1 + 0.1 * rn(D)
    beta = 0.1 * rn(D)
    _, mean, rstd = H.layernorm_fwd(x, gamma, beta, ROWS, D)
    W1 = rn(DFF, D, sc=0.02).to(BF)
    W2 = rn(D, DFF, sc=0.02).to(BF)
    Wqkv = rn(3 * D, D, sc=0.02).to(BF)
    dpre = rn(ROWS, DFF, sc=1e-3).to(BF)
    pre = rn(ROWS, DFF).to(BF)
    act = rn(ROWS, DFF).to(BF)
    qkv = rn(ROWS, 3 * D, sc=0.5).to(BF)
    bias = 0.1 * rn(DFF)
    attn_o, stats = H.attn_fwd(qkv, B, S, HEADS, size=SIZE)
    dattn = rn(ROWS, D, sc=1e-3).to(BF)
    torch.cuda.synchronize()

    def ln_bwd():
        d = H.DeferredReduce(dev)
        dx, dg, db, dxs = H.layernorm_bwd(dy, x, gamma, mean, rstd, ROWS, D, dres=dres, defer=d, dx_colsum=True)
        d.flush()
        return [dx, dg, db, dxs]

    def ln_bwd_plain():
        return list(H.layernorm_bwd(dy, x, gamma, mean, rstd, ROWS, D, dres=dres))

    def ln_fwd():
        return list(H.layernorm_fwd(x, gamma, beta, ROWS, D))

    def gemm_dx():          # dh2 = dpre . W1  (B k-strided)
        return [H.gemm(dpre, W1, ROWS, D, DFF, b_kstrided=True)]

    def gemm_dx_gelu():     # dpre = (dx3 . W2) * gelu'(pre), fused column sums
        d = H.DeferredReduce(dev)
        o, cs = H.gemm(dy, W2, ROWS, DFF, D, b_kstrided=True, epilogue=L.EPI_GELU_BWD, resid=pre, colsum_defer=d)
        d.flush()
        return [o, cs]

    def gemm_fwd_fc1():
        aux = torch.empty(ROWS, DFF, dtype=BF, device=dev)
        o = H.gemm(x, W1, ROWS, DFF, D, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)
        return [o, aux]

    def gemm_dw():
        from xpretrain_amd.functional import _wgrad
        return [_wgrad(dpre, x, ROWS, DFF, D)]

    def attn_bwd():
        return [H.attn_bwd(qkv, attn_o, dattn, stats, B, S, HEADS, size=SIZE, q_scale=0.125)]

    def attn_fwd():
        return list(H.attn_fwd(qkv, B, S, HEADS, size=SIZE))

    def pk_probe():
        err = torch.zeros(PK_VARIANTS * 128, dtype=torch.int32, device=dev)
        L.check(L.lib().xp_probe_pk_f32(err.data_ptr(), 2000, 2048, 12345, torch.cuda.current_stream().cuda_stream), "xp_probe_pk_f32")
        return [err]

    # the text tower's GEMMs (128x128 family, 64 KiB LDS: they CAN share a CU with the 52 KiB attention workgroups of the
    # video tower, unlike the 128 KiB 256x256 family) -- forward epilogues, dX, dW
    Rt, Dt, Dft = 256, 512, 2048
    xt = rn(Rt, Dt).to(BF); Wt1 = rn(Dft, Dt, sc=0.02).to(BF); Wt2 = rn(Dt, Dft, sc=0.02).to(BF); bt = 0.1 * rn(Dft)
    Wtq = rn(3 * Dt, Dt, sc=0.02).to(BF); btq = 0.1 * rn(3 * Dt); dyt = rn(Rt, Dft, sc=1e-3).to(BF); bt2 = 0.1 * rn(Dt)

    def text_gemms():
        aux = torch.empty(Rt, Dft, dtype=BF, device=dev)
        a = H.gemm(xt, Wt1, Rt, Dft, Dt, epilogue=L.EPI_BIAS_GELU, bias=bt, aux=aux)
        b = H.gemm(a, Wt2, Rt, Dt, Dft, epilogue=L.EPI_BIAS_RESID, bias=bt2, resid=xt)
        c = H.gemm(xt, Wtq, Rt, 3 * Dt, Dt, epilogue=L.EPI_BIAS_QSCALE, bias=btq, scale=0.125, scale_cols=Dt)
        d = H.gemm(dyt, Wt1, Rt, Dt, Dft, b_kstrided=True)
        e = H.gemm(dyt, xt, Dft, Dt, Rt, a_kstrided=True, b_kstrided=True, lda=Dft, ldb=Dt, out_dtype=torch.float32)
        return [a, aux, b, c, d, e]

    def gemm_then_ln():     # the producer / consumer pair as in the layer backward
        dh = H.gemm(dpre, W1, ROWS, D, DFF, b_kstrided=True)
        d = H.DeferredReduce(dev)
        dx, dg, db, dxs = H.layernorm_bwd(dh, x, gamma, mean, rstd, ROWS, D, dres=dres, defer=d, dx_colsum=True)
        d.flush()
        return [dh, dx, dg, db, dxs]

    return {k: v for k, v in locals().items() if callable(v) and k not in ("rn",)}


def make_aggressors(dev):
    g = torch.Generator(device=dev).manual_seed(11)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
    R, Dt, Dff = 256, 512, 2048
    x = rn(R, Dt).to(BF)
    dy = rn(R, Dt, sc=1e-3).to(BF)
    gamma, beta = 1 + 0.1 * rn(Dt), 0.1 * rn(Dt)
    _, mean, rstd = H.layernorm_fwd(x, gamma, beta, R, Dt)
    W = rn(Dff, Dt, sc=0.02).to(BF)
    dpre = rn(R, Dff, sc=1e-3).to(BF)
    qkv = rn(R, 3 * Dt, sc=0.5).to(BF)
    mask = torch.ones(8, 32, dtype=torch.int64, device=dev)
    mask[:, 20:] = 0
    ao, st = H.attn_fwd(qkv, 8, 32, 8, pad_mask=mask)
    big = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    big2 = torch.empty_like(big)
    torch.cuda.synchronize()

    def ln_text():
        H.layernorm_bwd(dy, x, gamma, mean, rstd, R, Dt, dres=dy)

    def ln_text_fwd():
        H.layernorm_fwd(x, gamma, beta, R, Dt)

    def gemm_text():
        H.gemm(dpre, W, R, Dt, Dff, b_kstrided=True)

    def gemm_text_noglds():
        os.environ["XPRETRAIN_DEBUG"] = "gemm_no_glds"          # (read by the library at every call: csrc/common.cpp::xp_debug_flag)
        try:
            H.gemm(dpre, W, R, Dt, Dff, b_kstrided=True)
        finally:
            del os.environ["XPRETRAIN_DEBUG"]

    dpre_s, W_s = rn(R, 128, sc=1e-3).to(BF), rn(128, Dt, sc=0.02).to(BF)
    dpre_l, W_l = rn(R, 16384, sc=1e-3).to(BF), rn(16384, Dt, sc=0.02).to(BF)
    W_l2 = rn(Dt, 16384, sc=0.02).to(BF)

    def gemm_text_k128():
        H.gemm(dpre_s, W_s, R, Dt, 128, b_kstrided=True)

    def gemm_text_k16384():
        H.gemm(dpre_l, W_l, R, Dt, 16384, b_kstrided=True)

    def gemm_text_slowepi():
        os.environ["XPRETRAIN_DEBUG"] = "gemm_slow_epi"
        try:
            H.gemm(dpre, W, R, Dt, Dff, b_kstrided=True)
        finally:
            del os.environ["XPRETRAIN_DEBUG"]

    def gemm_text_fwd_k16384():
        H.gemm(dpre_l, rn(Dt, 16384, sc=0.02).to(BF) if False else W_l2, R, Dt, 16384)

    def gemm_text_fwd():
        H.gemm(x, W, R, Dff, Dt)

    def gemm_text_dw():
        H.gemm(dpre, x, Dff, Dt, R, a_kstrided=True, b_kstrided=True, lda=Dff, ldb=Dt, out_dtype=torch.float32)

    def attn_text():
        H.attn_fwd(qkv, 8, 32, 8, pad_mask=mask)

    def attn_text_bwd():
        H.attn_bwd(qkv, ao, ao, st, 8, 32, 8, pad_mask=mask, q_scale=0.125)

    qkv_v = rn(ROWS, 3 * D, sc=0.5).to(BF)
    ao_v, st_v = H.attn_fwd(qkv_v, B, S, HEADS, size=SIZE)

    def attn_vision():           # 52 KiB LDS workgroups, MFMA: can share a CU with the 64 KiB 128x128 GEMM family
        H.attn_fwd(qkv_v, B, S, HEADS, size=SIZE)

    def attn_vision_bwd():
        H.attn_bwd(qkv_v, ao_v, ao_v, st_v, B, S, HEADS, size=SIZE, q_scale=0.125)

    def colsum_text():
        H.colsum(dpre, R, Dff)

    def torch_copy():
        big2.copy_(big)

    def torch_small():
        torch.add(dy, dy)

    def none():
        pass

    return {k: v for k, v in locals().items() if callable(v) and k not in ("rn",)}


PK_VARIANTS = 15


def run_pair(victim, agg, side, iters, agg_per_iter=4):
    """Run `victim` `iters` times on the current stream while `agg` loops on `side`; returns, per victim output, the number
    of runs whose result differs bitwise from the first run."""
    dev = torch.device("cuda", torch.cuda.current_device())
    ref = [t.clone() for t in victim()]
    torch.cuda.synchronize()
    bad = torch.zeros(len(ref), dtype=torch.int64, device=dev)
    for it in range(iters):
        with torch.cuda.stream(side):
            for _ in range(agg_per_iter):
                agg()
        if it % 50 == 0:          # poison every workspace of this stream
            for (d_, s_, tag), buf in list(H._ws_cache.items()):
                if s_ == torch.cuda.current_stream().cuda_stream:
                    buf.fill_(0xFF)
        outs = victim()
        for i, (o, r) in enumerate(zip(outs, ref)):
            bad[i] += (o != r).any().to(torch.int64)
            o.fill_(float("nan") if o.is_floating_point() else -1)      # the block returns to the allocator poisoned: the
        del outs                                                         # next run's torch.empty output starts as NaN
    torch.cuda.synchronize()
    return bad.tolist()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--victims", default="ln_bwd")
    ap.add_argument("--aggressors", default="none,ln_text,gemm_text,attn_text")
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--agg-per-iter", type=int, default=4)
    ap.add_argument("--analyze", default="", help="aggressor name: LayerNorm-backward event capture (see analyze())")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "determinism", "race_repro.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    V, A = make_victims(dev), make_aggressors(dev)
    side = torch.cuda.Stream(device=dev)
    res = {}
    for vn in a.victims.split(","):
        victim = V[vn]
        for an in a.aggressors.split(","):
            agg = A[an]
            bad = run_pair(victim, agg, side, a.iters, a.agg_per_iter)
            res[f"{vn}|{an}"] = bad
            if vn == "pk_probe":          # one more run, printed per variant / lane / half
                with torch.cuda.stream(side):
                    for _ in range(a.agg_per_iter * 4):
                        agg()
                e = victim()[0].view(PK_VARIANTS, 64, 2).cpu()
                for v in range(PK_VARIANTS):
                    print(f"   pk variant {v}: errors lo-half {int(e[v, :, 0].sum())} hi-half {int(e[v, :, 1].sum())}; by 16-lane group lo "
                          f"{[int(e[v, g * 16:(g + 1) * 16, 0].sum()) for g in range(4)]} hi {[int(e[v, g * 16:(g + 1) * 16, 1].sum()) for g in range(4)]}")
            print(f"victim {vn:14s} aggressor {an:14s}: outputs differing from the first run in {bad} of {a.iters} runs", flush=True)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    if a.analyze:
        analyze(dev, A[a.analyze], side, a.iters, a.agg_per_iter, a.out.replace(".json", "_events.pt"))


def analyze(dev, agg, side, iters, agg_per_iter, out):
    """LayerNorm backward as victim: for every run keep (first differing row, #differing rows, that row of dx, dgamma,
    dxs); afterwards dump the events with the inputs of the affected rows so the perturbation can be solved for offline."""
    g = torch.Generator(device=dev).manual_seed(7)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
    x = rn(ROWS, D).to(BF)
    dy = rn(ROWS, D, sc=1e-3).to(BF)
    dres = rn(ROWS, D, sc=1e-3).to(BF)
    gamma = 1 + 0.1 * rn(D)
    beta = 0.1 * rn(D)
    _, mean, rstd = H.layernorm_fwd(x, gamma, beta, ROWS, D)

    def victim():
        d = H.DeferredReduce(dev)
        r = H.layernorm_bwd(dy, x, gamma, mean, rstd, ROWS, D, dres=dres, defer=d, dx_colsum=True)
        d.flush()
        return r
    ref = [t.clone() for t in victim()]
    torch.cuda.synchronize()
    row = torch.zeros(iters, dtype=torch.int64, device=dev)
    nrow = torch.zeros(iters, dtype=torch.int64, device=dev)
    ev_dx = torch.zeros(iters, D, dtype=BF, device=dev)
    ev_dg = torch.zeros(iters, D, device=dev)
    ev_db = torch.zeros(iters, D, device=dev)
    ev_dxs = torch.zeros(iters, D, device=dev)
    for it in range(iters):
        with torch.cuda.stream(side):
            for _ in range(agg_per_iter):
                agg()
        dx, dg, db, dxs = victim()
        rd = (dx != ref[0]).any(dim=1)
        nrow[it] = rd.sum()
        r = rd.to(torch.int8).argmax()
        row[it] = r
        ev_dx[it] = dx.index_select(0, r.reshape(1))[0]
        ev_dg[it], ev_db[it], ev_dxs[it] = dg, db, dxs
    torch.cuda.synchronize()
    idx = (nrow > 0).nonzero().flatten()
    rows = row[idx]
    print(f"analyze: {idx.numel()} events in {iters} runs; rows {rows.tolist()[:20]}; rows differing per event {nrow[idx].tolist()[:20]}")
    torch.save({"iters": idx.cpu(), "rows": rows.cpu(), "nrow": nrow[idx].cpu(), "ev_dx": ev_dx[idx].cpu(), "ev_dg": ev_dg[idx].cpu(),
                "ev_db": ev_db[idx].cpu(), "ev_dxs": ev_dxs[idx].cpu(), "ref_dx": ref[0][rows].cpu(), "ref_dg": ref[1].cpu(),
                "ref_db": ref[2].cpu(), "ref_dxs": ref[3].cpu(), "x": x[rows].cpu(), "dy": dy[rows].cpu(), "dres": dres[rows].cpu(),
                "mean": mean[rows].cpu(), "rstd": rstd[rows].cpu(), "gamma": gamma.cpu()}, out)


if __name__ == "__main__":
    main()
