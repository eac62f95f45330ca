#!/usr/bin/env python
"""Micro-benchmarks of the individual HIP kernels at BASELINE cfg #2 shapes (B=8, T=12, 224^2, ViT-B/16).
Run on the GPU box:  python tools/bench_kernels.py [gemm|gemmfwd|ln|attn|all]  -> prints one line per kernel."""
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H  # noqa: E402
from xpretrain_amd import _lib as L  # noqa: E402


def timeit(fn, iters=200, warmup=300):     # (sustained clocks: a cold GPU ramps for the first few hundred launches, tools/power_probe.py)
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters * 1e3   # us


def bench_gemm(fwd_only=False):
    M = 8 * 2356
    bf = torch.bfloat16
    for name, N, K, kw in [("qkv", 2304, 768, dict(epilogue=L.EPI_BIAS_QSCALE, scale=0.125, scale_cols=768)),
                           ("out", 768, 768, dict(epilogue=L.EPI_BIAS_RESID)),
                           ("fc1", 3072, 768, dict(epilogue=L.EPI_BIAS_GELU)),
                           ("fc2", 768, 3072, dict(epilogue=L.EPI_BIAS_RESID))]:
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") * 0.02).to(bf)
        bias = torch.zeros(N, device="cuda")
        out = torch.empty(M, N, dtype=bf, device="cuda")
        if kw["epilogue"] == L.EPI_BIAS_RESID:
            kw["resid"] = torch.randn(M, N, device="cuda").to(bf)
        if kw["epilogue"] == L.EPI_BIAS_GELU:
            kw["aux"] = torch.empty(M, N, dtype=bf, device="cuda")
        us = timeit(lambda: H.gemm(A, W, M, N, K, out=out, bias=bias, **kw))
        print(f"gemm fwd {name:4s} M={M} N={N} K={K}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s")
        if fwd_only:
            continue
        # dX: dY[M,N] . W[N,K]
        dY = torch.randn(M, N, device="cuda").to(bf)
        dX = torch.empty(M, K, dtype=bf, device="cuda")
        us = timeit(lambda: H.gemm(dY, W, M, K, N, b_kstrided=True, out=dX))
        print(f"gemm dX  {name:4s}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s")
        # dW[N,K] = dY^T X: the production path (functional._wgrad: split-K chosen by xp_gemm_auto_split + deterministic reduce)
        from xpretrain_amd.functional import _wgrad, _split_for
        split = _split_for(N, K, M, bf, (0, 0, 0))        # (the general plan; fc2 / fc1 / out_proj run the "slack" plan inside a layer's backward)
        us = timeit(lambda: _wgrad(dY, A, M, N, K))
        print(f"gemm dW  {name:4s} split={split:2d} (auto): {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s")
        us = timeit(lambda: H.colsum(dY, M, N))
        print(f"colsum   {name:4s}: {us:8.1f} us  {M*N*2/us/1e3:7.1f} GB/s")


def bench_ln():
    M, D = 8 * 2356, 768
    x = torch.randn(M, D, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    us = timeit(lambda: H.layernorm_fwd(x, g, b, M, D))
    print(f"layernorm fwd {M}x{D}: {us:7.1f} us  {M*D*4/us/1e3:7.1f} GB/s")
    y, mean, rstd = H.layernorm_fwd(x, g, b, M, D)
    dy = torch.randn_like(x)
    us = timeit(lambda: H.layernorm_bwd(dy, x, g, mean, rstd, M, D, dres=dy))
    print(f"layernorm bwd {M}x{D}: {us:7.1f} us  {M*D*8/us/1e3:7.1f} GB/s")


def bench_attn():
    B, Hh, M, N, Lp = 8, 12, 4, 12, 196
    S = M + N * Lp
    qkv = torch.randn(B * S, 3 * Hh * 64, device="cuda").to(torch.bfloat16)
    us = timeit(lambda: H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp)))
    flops = 4 * N * Lp * (M + Lp) * 64 * Hh * B + 4 * M * S * 64 * Hh * B
    print(f"attn fwd  B{B} H{Hh} (4,12,196): {us:7.1f} us  {flops/us/1e6:6.1f} TFLOP/s  {4*B*S*Hh*64*2/us/1e3:7.1f} GB/s")
    out, stats = H.attn_fwd(qkv, B, S, Hh, size=(M, N, Lp))
    dout = torch.randn_like(out)
    us = timeit(lambda: H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=(M, N, Lp), q_scale=0.125))
    print(f"attn bwd  B{B} H{Hh} (4,12,196): {us:7.1f} us  {2.5*flops/us/1e6:6.1f} TFLOP/s")
    ids_mask = torch.ones(8, 32, dtype=torch.int64, device="cuda")
    q2 = torch.randn(8 * 32, 3 * 8 * 64, device="cuda").to(torch.bfloat16)
    us = timeit(lambda: H.attn_fwd(q2, 8, 32, 8, pad_mask=ids_mask))
    print(f"attn fwd  text B8 H8 S32: {us:7.1f} us")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what == "gemmfwd":
        bench_gemm(fwd_only=True)
    if what in ("ln", "all"):
        bench_ln()
    if what in ("attn", "all"):
        bench_attn()
