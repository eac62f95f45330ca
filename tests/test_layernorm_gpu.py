"""GPU: LayerNorm fwd/bwd kernels vs torch fp64 on the kernel's own (rounded) inputs."""
import pytest
import torch

from tests.gpu_util import report

pytestmark = pytest.mark.gpu
TOL = {torch.bfloat16: 6e-3, torch.float32: 1e-5}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(204, 768), (7, 512), (1030, 128), (33, 1024), (5, 192)])
def test_layernorm_fwd_bwd(dtype, rows, cols):
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(rows + cols)
    x = (torch.randn(rows, cols, device="cuda") * 2 + 0.5).to(dtype)
    g = torch.randn(cols, device="cuda") * 0.3 + 1
    b = torch.randn(cols, device="cuda") * 0.1
    y, mean, rstd = H.layernorm_fwd(x, g, b, rows, cols)
    xd = x.double().requires_grad_()
    gd, bd = g.double().requires_grad_(), b.double().requires_grad_()
    yref = torch.nn.functional.layer_norm(xd, (cols,), gd, bd, 1e-5)
    tol = TOL[dtype]
    assert report(f"ln_fwd {dtype} {rows}x{cols}", y, yref, tol) <= tol
    assert report("ln_mean", mean, xd.mean(-1), 1e-5) <= 1e-5
    dy = torch.randn(rows, cols, device="cuda").to(dtype)
    dres = torch.randn(rows, cols, device="cuda").to(dtype)
    yref.backward(dy.double())
    dx, dg, db = H.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dres=dres)
    assert report(f"ln_dx {dtype}", dx, xd.grad + dres.double(), tol) <= tol
    assert report(f"ln_dgamma {dtype}", dg, gd.grad, 1e-4) <= 1e-4
    assert report(f"ln_dbeta {dtype}", db, bd.grad, 1e-4) <= 1e-4
    # accumulate + no residual
    dx2, dg2, db2 = H.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dgamma=dg.clone(), dbeta=db.clone(), accumulate=True)
    assert report(f"ln_dx_nores {dtype}", dx2, xd.grad, tol) <= tol
    assert report(f"ln_dgamma_acc {dtype}", dg2, 2 * gd.grad, 1e-4) <= 1e-4


def test_layernorm_strided_rows():
    """pooled rows: x[:, 0] of a [B,S,D] tensor read in place through ldx = S*D."""
    from xpretrain_amd import hip_ops as H
    B, S, D = 6, 11, 256
    x = torch.randn(B, S, D, device="cuda").to(torch.bfloat16)
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    y, _, _ = H.layernorm_fwd(x, g, b, B, D, ldx=S * D)
    ref = torch.nn.functional.layer_norm(x[:, 0].double(), (D,))
    assert report("ln_strided", y, ref, 6e-3) <= 6e-3


def test_reduce_rows_batch_and_deferred_producers():
    """xp_reduce_rows_batch: many segments, direct (<= 64 rows) and two-level, pitch > width, accumulate; and the
    deferred forms of colsum / layernorm_bwd give the immediate forms' results (bit-identical reduction trees are
    not required -- fp32 roundoff tolerance)."""
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(4)
    d = H.DeferredReduce(torch.device("cuda"))
    cases, outs, refs = [(1, 64, 64), (64, 200, 256), (65, 768, 768), (589, 3072, 3072), (1024, 768, 1536), (300, 4, 8)], [], []
    for nrows, width, stride in cases:
        part = torch.randn(nrows, stride, device="cuda")
        out = torch.full((width,), 3.0, device="cuda")
        acc = nrows % 2 == 0
        d.add(part, 0, out, nrows, width, stride, accumulate=acc)
        outs.append(out)
        refs.append(part[:, :width].double().sum(0) + (3.0 if acc else 0.0))
    d.flush()
    for (nrows, width, stride), o, r in zip(cases, outs, refs):
        assert report(f"reduce_batch {nrows}x{width}/{stride}", o, r, 2e-6) <= 2e-6

    # four columns per lane (every segment 16-byte addressable: the encoder layers' segments) against one column per lane
    # (XPRETRAIN_DEBUG=rows_reduce_scalar; also what a batch with an odd width / pitch falls back to): the same summation order per column
    import os
    cases4 = [(1, 64, 64), (64, 200, 256), (65, 768, 768), (589, 3072, 3072), (1024, 768, 1536), (512, 768, 3072), (148, 3072, 3072), (300, 4, 8)]
    parts = [torch.randn(nrows, stride, device="cuda") for nrows, _, stride in cases4]
    saved = os.environ.get("XPRETRAIN_DEBUG")

    def run(scalar):
        os.environ["XPRETRAIN_DEBUG"] = ",".join(filter(None, [saved, "rows_reduce_scalar" if scalar else ""]))
        try:
            dd, res = H.DeferredReduce(torch.device("cuda")), []
            for (nrows, width, stride), part in zip(cases4, parts):
                out = torch.full((width,), 3.0, device="cuda")
                dd.add(part, 0, out, nrows, width, stride, accumulate=nrows % 2 == 0)
                res.append(out)
            dd.flush()
            torch.cuda.synchronize()
            return res
        finally:
            if saved is None:
                os.environ.pop("XPRETRAIN_DEBUG", None)
            else:
                os.environ["XPRETRAIN_DEBUG"] = saved
    for (nrows, width, stride), a, b in zip(cases4, run(False), run(True)):
        assert torch.equal(a, b), (nrows, width, stride)

    for dtype in (torch.bfloat16, torch.float32):
        rows, cols = 2356, 768
        X = torch.randn(rows, cols, device="cuda").to(dtype)
        dy, x = torch.randn(rows, cols, device="cuda").to(dtype), torch.randn(rows, cols, device="cuda").to(dtype)
        gamma = torch.randn(cols, device="cuda")
        mean, rstd = x.float().mean(1), 1.0 / x.float().var(1, unbiased=False).add(1e-5).sqrt()
        d = H.DeferredReduce(X.device)
        cs = H.colsum_deferred(X, rows, cols, d)
        dx1, dg1, db1 = H.layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols, defer=d, name="ln_a")
        dres = torch.randn(rows, cols, device="cuda").to(dtype)
        dx2, dg2, db2, dxs = H.layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols, dres=dres, defer=d, dx_colsum=True, name="ln_b")
        # 4-vector partials: additionally the column sums of the residual gradient the kernel reads (fc2's bias gradient)
        dx3, dg3, db3, dxs3, drs3 = H.layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols, dres=dres, defer=d, dx_colsum=True,
                                                    dres_colsum=True, name="ln_c")
        with pytest.raises(RuntimeError, match="two producers"):      # slots are named: a second producer of a name cannot alias
            H.colsum_deferred(X, rows, cols, d)
        d.flush()
        assert torch.equal(dx3, dx2)
        assert report(f"ln dres colsum {dtype}", drs3, dres.double().sum(0), 1e-5) <= 1e-5
        assert report(f"ln dx colsum (4-vector partials) {dtype}", dxs3, dxs, 2e-6) <= 2e-6
        assert report(f"ln dgamma (4-vector partials) {dtype}", dg3, dg2, 2e-6) <= 2e-6
        assert report(f"ln dx colsum {dtype}", dxs, dx2.double().sum(0), 3e-3 if dtype == torch.bfloat16 else 1e-5) <= 3e-3
        assert report(f"ln dgamma (3-vector partials) {dtype}", dg2, dg1, 2e-6) <= 2e-6
        dx0, dg0, db0 = H.layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols)
        assert report(f"colsum deferred {dtype}", cs, H.colsum(X, rows, cols), 2e-6) <= 2e-6
        assert torch.equal(dx0, dx1)
        assert report(f"ln dgamma deferred {dtype}", dg1, dg0, 2e-6) <= 2e-6
        assert report(f"ln dbeta deferred {dtype}", db1, db0, 2e-6) <= 2e-6


@pytest.mark.parametrize("rows,S,M,stride", [(2 * 2356, 2356, 4, 4), (8, 1, 1, 4), (60, 10, 3, 5)])
def test_forward_with_fp32_side_rows(rows, S, M, stride):
    """xp_layernorm_fwd_side: rows r with r % S < M are read from the fp32 side buffer (row (r // S) * stride + r % S) instead of the
    bf16 x, and -- on request -- their fp32 result is written to a second side buffer (pre_layrnorm: its output is the stream)."""
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(rows)
    cols = 768
    x = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    nb = (rows + S - 1) // S
    xs = torch.randn(nb * stride, cols, device="cuda") * 1.5 + 0.3
    ys = torch.full_like(xs, float("nan"))
    g, b = torch.randn(cols, device="cuda"), torch.randn(cols, device="cuda")
    y, mean, rstd = H.layernorm_fwd(x, g, b, rows, cols, x_side=xs, y_side=ys, side=(S, M, stride))
    y0, mean0, rstd0 = H.layernorm_fwd(x, g, b, rows, cols)
    r = torch.arange(rows, device="cuda")
    is_side, sidx = (r % S) < M, (r // S) * stride + r % S
    assert torch.equal(y[~is_side], y0[~is_side]) and torch.equal(mean[~is_side], mean0[~is_side])
    ref = torch.nn.functional.layer_norm(xs[sidx[is_side]].double(), (cols,), g.double(), b.double(), 1e-5)
    assert report("ln side rows fp32 out", ys[sidx[is_side]], ref, 2e-5) <= 2e-5
    assert torch.equal(y[is_side], ys[sidx[is_side]].to(torch.bfloat16))
    assert report("ln side rows mean", mean[is_side], xs[sidx[is_side]].double().mean(1), 1e-5, scale_floor=1e-2) <= 1e-5
    touched = torch.zeros(nb * stride, dtype=torch.bool, device="cuda"); touched[sidx[is_side]] = True
    assert bool(torch.isnan(ys[~touched]).all())
    y1, _, _ = H.layernorm_fwd(x, g, b, rows, cols, x_side=xs, side=(S, M, stride))       # read-only form
    assert torch.equal(y1, y)


@pytest.mark.parametrize("rows,S,M,stride", [(2 * 2356, 2356, 4, 4), (8, 1, 1, 1), (60, 10, 3, 5)])
def test_backward_with_fp32_side_rows(rows, S, M, stride):
    """xp_layernorm_bwd_side / _partials_side: the rows the forward normalised from the fp32 side buffer get their x-hat from the same
    fp32 values in the backward (fp64 autograd of the mixed-precision input as reference); the other rows are bit-identical to the
    plain kernel; both launch forms (immediate and deferred partial rows) agree."""
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(rows + 1)
    cols = 768
    x = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    nb = (rows + S - 1) // S
    xs = torch.randn(nb * stride, cols, device="cuda") * 1.5 + 0.3
    g, b = torch.randn(cols, device="cuda"), torch.randn(cols, device="cuda")
    dy = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    dres = torch.randn(rows, cols, device="cuda").to(torch.bfloat16)
    side = (S, M, stride)
    _, mean, rstd = H.layernorm_fwd(x, g, b, rows, cols, x_side=xs, side=side)
    dx, dg, db = H.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dres=dres, x_side=xs, side=side)
    dx0, dg0, db0 = H.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dres=dres)
    r = torch.arange(rows, device="cuda")
    is_side, sidx = (r % S) < M, (r // S) * stride + r % S
    assert torch.equal(dx[~is_side], dx0[~is_side])
    assert not torch.equal(dx[is_side], dx0[is_side])           # the bf16 x of those rows is unrelated noise here
    xin = x.double()
    xin[is_side] = xs[sidx[is_side]].double()
    xin.requires_grad_(True)
    gd = g.double().requires_grad_(True)
    bd = b.double().requires_grad_(True)
    torch.nn.functional.layer_norm(xin, (cols,), gd, bd, 1e-5).backward(dy.double())
    assert report("ln bwd side rows dx", dx[is_side], xin.grad[is_side] + dres[is_side].double(), 1.5e-2) <= 1.5e-2   # bf16 output
    assert report("ln bwd side dgamma", dg, gd.grad, 2e-5) <= 2e-5
    assert report("ln bwd side dbeta", db, bd.grad, 2e-5) <= 2e-5
    d = H.DeferredReduce(x.device)
    dx1, dg1, db1, dxs, drs = H.layernorm_bwd(dy, x, g, mean, rstd, rows, cols, dres=dres, defer=d, dx_colsum=True, dres_colsum=True,
                                              name="ln_side", x_side=xs, side=side)
    d.flush()
    assert torch.equal(dx1, dx)
    assert report("ln bwd side dgamma (deferred)", dg1, dg, 2e-6) <= 2e-6
    assert report("ln bwd side dx colsum", dxs, dx.double().sum(0), 3e-3) <= 3e-3

