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
