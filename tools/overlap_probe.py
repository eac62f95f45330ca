#!/usr/bin/env python
"""Can the HBM-bound reductions of a layer backward (split-K reduce, bias column sums) hide beside the MFMA-bound GEMMs when
they are issued on a second stream?  Times main-stream GEMMs alone, the side-stream reductions alone, and both together."""
import sys

import torch

sys.path.insert(0, ".")
from xpretrain_amd import hip_ops as H, _lib as L  # noqa: E402

M, D, Dff = 8 * 2356, 768, 3072
bf = torch.bfloat16
dev = "cuda"
dpre = (torch.randn(M, Dff, device=dev) * 1e-3).to(bf)
W1 = (torch.randn(Dff, D, device=dev) * 0.02).to(bf)
h2 = torch.randn(M, D, device=dev).to(bf)
dh2 = torch.empty(M, D, dtype=bf, device=dev)
slabs_a = torch.empty(7, Dff, D, device=dev)
slabs_b = torch.randn(7, Dff, D, device=dev)
dw = torch.empty(Dff, D, device=dev)
dqkv = torch.randn(M, 3 * D, device=dev).to(bf)
part = torch.empty(H.L.lib().xp_colsum_partial_rows(M, 3 * D) * 3 * D, device=dev)
x = torch.randn(M, D, device=dev).to(bf)
Wf = (torch.randn(Dff, D, device=dev) * 0.02).to(bf)
bias = torch.zeros(Dff, device=dev)
out = torch.empty(M, Dff, dtype=bf, device=dev)
aux = torch.empty(M, Dff, dtype=bf, device=dev)
side = torch.cuda.Stream()


def gemms(which):
    if which == "dx":      # 222 tiles, K = 3072
        H.gemm(dpre, W1, M, D, Dff, b_kstrided=True, out=dh2)
    elif which == "dw":    # 252 workgroups
        H.gemm(dpre, h2, Dff, D, M, a_kstrided=True, b_kstrided=True, lda=Dff, ldb=D, split_k=7, out=slabs_a)
    else:                  # 888 tiles
        H.gemm(x, Wf, M, Dff, D, out=out, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)


def reductions():
    H.splitk_reduce(slabs_b, dw, splits=7)
    L.check(L.lib().xp_colsum_partials(dqkv.data_ptr(), M, 3 * D, 3 * D, L.XP_BF16, part.data_ptr(), part.numel() * 4,
                                        torch.cuda.current_stream().cuda_stream), "colsum")


def timed(f, n=40):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    side_done = torch.cuda.Event(); side_done.record(side)
    torch.cuda.current_stream().wait_event(side_done)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def both(which):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            reductions()
        gemms(which)
    return f


t_red = timed(reductions)
for which in ("dx", "dw", "fwd"):
    t_g = timed(lambda: gemms(which))
    t_b = timed(both(which))
    print(f"{which:3s}: GEMM alone {t_g:6.1f} us, reductions alone {t_red:5.1f} us, serial {t_g + t_red:6.1f} us, two streams {t_b:6.1f} us "
          f"-> hidden {t_g + t_red - t_b:5.1f} us")
