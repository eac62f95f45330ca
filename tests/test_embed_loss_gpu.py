"""GPU: embedding / pooling / normalise / cast kernels and the fused NCE loss vs torch fp64 and the
reference-generated loss fixtures (tests/golden/loss.pt)."""
import math

import pytest
import torch

from tests.gpu_util import report

pytestmark = pytest.mark.gpu
DTYPES = [torch.bfloat16, torch.float32]
TOL = {torch.bfloat16: 6e-3, torch.float32: 1e-5}


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("P,R", [(16, 64), (8, 32), (32, 64)])
def test_im2col(dtype, P, R):
    from xpretrain_amd import hip_ops as H
    v = torch.randn(5, 3, R, R, device="cuda")
    got = H.im2col(v, P, dtype)
    g = R // P
    ref = v.view(5, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(5 * g * g, 3 * P * P)
    assert torch.equal(got.float(), ref.to(dtype).float())


@pytest.mark.parametrize("dtype", DTYPES)
def test_vip_rows_and_embed_bwd(dtype):
    from xpretrain_amd import hip_ops as H
    B, M, T, Lp, D = 3, 4, 5, 9, 128
    S = M + T * Lp
    cls, add, pos = torch.randn(D, device="cuda"), torch.randn(M - 1, D, device="cuda"), torch.randn(1 + Lp, D, device="cuda")
    x = torch.zeros(B * S, D, dtype=dtype, device="cuda")
    H.vip_proxy_rows(cls, add, pos, x, B, S, M, D)
    ref = torch.zeros(B, S, D, device="cuda")
    ref[:, 0] = cls + pos[0]; ref[:, 1:M] = add + pos[0]
    assert torch.equal(x.view(B, S, D).float(), ref.to(dtype).float())
    dx = torch.randn(B, S, D, device="cuda").to(dtype)
    dc, da, dp, dt = H.vip_embed_bwd(dx, B, M, T, Lp, D)
    d = dx.double()
    fr = d[:, M:].view(B, T, Lp, D)
    assert report("d_class", dc, d[:, 0].sum(0), 1e-5) <= 1e-5
    assert report("d_added", da, d[:, 1:M].sum(0), 1e-5) <= 1e-5
    assert report("d_pos0", dp[0], d[:, :M].sum((0, 1)), 1e-5) <= 1e-5
    assert report("d_pos", dp[1:], fr.sum((0, 1)), 1e-5) <= 1e-5
    assert report("d_time", dt, fr.sum((0, 2)), 1e-5) <= 1e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_text_embed_and_pooling(dtype):
    from xpretrain_amd import hip_ops as H
    B, Lt, D, V = 4, 12, 128, 50
    ids = torch.randint(0, V - 1, (B, Lt), device="cuda")
    ids[:, 7:] = V - 1                      # repeated EOT: duplicate indices + first-max argmax
    ids[0, 3:] = V - 1
    tok, pos = torch.randn(V, D, device="cuda"), torch.randn(16, D, device="cuda")
    x = H.text_embed_fwd(ids, tok, pos, dtype)
    ref = tok[ids] + pos[:Lt][None]
    assert torch.equal(x.view(B, Lt, D).float(), ref.to(dtype).float())
    idx = H.argmax_rows(ids)
    assert torch.equal(idx.cpu(), ids.cpu().argmax(-1))
    pooled = H.gather_rows(x, idx, B, Lt, D)
    assert torch.equal(pooled, x.view(B, Lt, D)[torch.arange(B), idx])
    back = H.scatter_rows(pooled, idx, B, Lt, D).view(B, Lt, D)
    exp = torch.zeros_like(back); exp[torch.arange(B), idx] = pooled
    assert torch.equal(back, exp)
    first = H.gather_rows(x, None, B, Lt, D)
    assert torch.equal(first, x.view(B, Lt, D)[:, 0])
    dx = torch.randn(B, Lt, D, device="cuda").to(dtype)
    d_tok, d_pos = H.text_embed_bwd(ids, dx, V, 16)
    rt = torch.zeros(V, D, dtype=torch.float64, device="cuda").index_add_(0, ids.view(-1), dx.double().view(-1, D))
    assert report("d_tok", d_tok, rt, 1e-5) <= 1e-5
    assert report("d_pos_text", d_pos[:Lt], dx.double().sum(0), 1e-5) <= 1e-5
    assert d_pos[Lt:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_l2norm_and_cast(dtype):
    from xpretrain_amd import hip_ops as H
    x = torch.randn(9, 512, device="cuda").to(dtype)
    y, inv = H.l2norm_fwd(x, 9, 512)
    xd = x.double().requires_grad_()
    yr = xd / xd.norm(dim=-1, keepdim=True)
    assert report("l2norm_fwd", y, yr, 1e-5) <= 1e-5
    dy = torch.randn(9, 512, device="cuda")
    yr.backward(dy.double())
    dx = H.l2norm_bwd(dy, y, inv, 9, 512, dtype)
    assert report("l2norm_bwd", dx, xd.grad, TOL[dtype]) <= TOL[dtype]
    w = torch.randn(1000, 12, device="cuda")
    assert torch.equal(H.cast(w, dtype).float(), w.to(dtype).float())
    assert torch.equal(H.cast_back(H.cast(w, dtype)), w.to(dtype).float())


def test_nce_loss_against_reference_fixtures(golden):
    """fp32 kernel vs outputs of the reference's NCELearnableTempLoss (tests/golden/loss.pt)."""
    from xpretrain_amd import hip_ops as H
    for c in golden("loss.pt"):
        v, t = c["feats"][0].cuda(), c["feats"][1].cuda()
        ls = torch.tensor(c["log_scale"], device="cuda")
        loss, dv, dt, dls = H.nce_loss(v, t, ls)
        tag = f"nce n={c['n']} ls={c['log_scale']:.2f}"
        assert abs(loss.item() - c["nce"].item()) <= 1e-3 * max(1.0, abs(c["nce"].item())), tag
        assert report(tag + " dV", dv, c["nce_grads"][0], 1e-3) <= 1e-3
        assert report(tag + " dT", dt, c["nce_grads"][1], 1e-3) <= 1e-3
        assert abs(dls.item() - c["nce_grads"][2].item()) <= 1e-3 * max(1.0, abs(c["nce_grads"][2].item())), tag


def test_nce_loss_large_n_fp64():
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(0)
    n, d = 200, 512
    v = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=-1)
    t = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=-1)
    ls = torch.tensor(4.6, device="cuda")
    loss, dv, dt, dls = H.nce_loss(v, t, ls)
    vd, td, lsd = v.double().requires_grad_(), t.double().requires_grad_(), ls.double().requires_grad_()
    A = vd @ td.t() * lsd.exp()
    lbl = torch.arange(n, device="cuda")
    ref = torch.nn.functional.cross_entropy(A, lbl) + torch.nn.functional.cross_entropy(A.t(), lbl)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-4
    assert report("nce200 dV", dv, vd.grad, 1e-4) <= 1e-4
    assert report("nce200 dT", dt, td.grad, 1e-4) <= 1e-4
    assert abs(dls.item() - lsd.grad.item()) < 1e-3


def test_vsc_fc_loss_against_reference_fixtures(golden):
    """fp32 kernel vs outputs of the reference's NCELearnableTempLoss_vsc_fc (tests/golden/loss.pt), through the module
    surface (autograd): loss, d{vis, txt, img, cap}, d log_scale."""
    from xpretrain_amd.optimization import build_loss_func
    fn = build_loss_func({"loss_name": "NCELearnableTempLoss_vsc_fc"})
    for c in golden("loss.pt"):
        feats = [f.cuda().requires_grad_() for f in c["feats"]]
        ls = torch.tensor(c["log_scale"], device="cuda", requires_grad=True)
        loss = fn(feats[0], feats[1], feats[2], feats[3], ls)
        grads = torch.autograd.grad(loss * 2.0, feats + [ls])            # incoming scalar 2.0 exercises backward scaling
        tag = f"vsc_fc n={c['n']} ls={c['log_scale']:.2f}"
        assert abs(loss.item() - c["vsc_fc"].item()) <= 1e-3 * max(1.0, abs(c["vsc_fc"].item())), tag
        for name, g, r in zip(("dV", "dT", "dI", "dC"), grads[:4], c["vsc_fc_grads"][:4]):
            assert report(f"{tag} {name}", g / 2.0, r, 1e-3) <= 1e-3
        r = c["vsc_fc_grads"][4].item()
        assert abs(grads[4].item() / 2.0 - r) <= 1e-3 * max(1.0, abs(r)), tag


def test_vsc_fc_loss_large_n_fp64():
    """n=200 (> one wave, > one sgemm tile) against the fp64 oracle."""
    from oracle import clipvip_oracle as O
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(21)
    n, d = 200, 96
    feats = [torch.nn.functional.normalize(torch.randn(n, d, dtype=torch.float64), dim=-1).requires_grad_() for _ in range(4)]
    ls = torch.tensor(3.7, dtype=torch.float64, requires_grad=True)
    ref = O.nce_vsc_fc_loss(*feats, ls)
    rg = torch.autograd.grad(ref, feats + [ls])
    out = H.vsc_fc_loss(*[f.detach().float().cuda() for f in feats], ls.detach().float().cuda())
    assert abs(out[0].item() - ref.item()) <= 1e-4 * abs(ref.item())
    for name, g, r in zip(("dV", "dT", "dI", "dC"), out[1:5], rg[:4]):
        assert report(f"vsc_fc200 {name}", g, r, 1e-4) <= 1e-4
    assert abs(out[5].item() - rg[4].item()) <= 1e-4 * max(1.0, abs(rg[4].item()))


@pytest.mark.parametrize("u8", [False, True])
@pytest.mark.parametrize("P,R,BT", [(16, 224, 5), (32, 224, 3), (16, 64, 7), (8, 40, 2)])
def test_patch_gather_in_the_gemm_loader(u8, P, R, BT):
    """XpGemmDesc::a_frames: the conv-as-GEMM of the patch embedding (CLIP_ViP.py:157-159,178) with the patch matrix gathered by
    the operand loader straight from [BT,3,H,W] (fp32, or decoded uint8 with the collate arithmetic fused in) -- bit-identical to
    the materialised im2col matrix through the same GEMM, plain and with the embedding epilogue (+temporal +position, token slots),
    ragged last tile included."""
    from xpretrain_amd import hip_ops as H
    from xpretrain_amd import _lib as L
    torch.manual_seed(P + R + BT)
    bf = torch.bfloat16
    g = R // P
    Lp, K, D = g * g, 3 * P * P, 256
    if u8:
        frames = torch.randint(0, 256, (BT, 3, R, R), dtype=torch.uint8, device="cuda")
        patches = H.im2col_u8(frames, P, bf)
        kw = dict(frames=frames, frame_patch=P, frame_norm=(H.CLIP_MEAN, H.CLIP_STD))
    else:
        frames = torch.randn(BT, 3, R, R, device="cuda")
        patches = H.im2col(frames, P, bf)
        kw = dict(frames=frames, frame_patch=P)
    W = (torch.randn(D, K, device="cuda") * 0.05).to(bf)
    M = BT * Lp
    want = H.gemm(patches, W, M, D, K)
    got = H.gemm(None, W, M, D, K, **kw)
    assert torch.equal(got, want)
    ref = patches.double() @ W.double().t()
    assert report(f"patch gather P{P} R{R} u8={u8}", got, ref, 6e-3) <= 6e-3
    # the embedding epilogue: + temporal[t] + position[1 + l], written to token slot Mp + t*L + l of sample b
    T, Mp = 1, 4
    Bv = BT // T
    S = Mp + T * Lp
    tt, pos = torch.randn(T, D, device="cuda"), torch.randn(Lp, D, device="cuda")
    x0 = torch.zeros(Bv * S, D, dtype=bf, device="cuda"); x1 = torch.zeros_like(x0)
    H.gemm(patches, W, M, D, K, out=x0, epilogue=L.EPI_PATCH, tab1=tt, tab2=pos, tab_L=Lp, c_remap=(T * Lp, S, Mp))
    H.gemm(None, W, M, D, K, out=x1, epilogue=L.EPI_PATCH, tab1=tt, tab2=pos, tab_L=Lp, c_remap=(T * Lp, S, Mp), **kw)
    assert torch.equal(x0, x1)


def test_patch_gather_rejects_what_it_cannot_do():
    from xpretrain_amd import hip_ops as H
    frames = torch.randn(2, 3, 32, 32, device="cuda")
    W = torch.zeros(64, 3 * 16 * 16, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="a_frames"):
        H.gemm(None, W, 2 * 4, 64, 3 * 16 * 16 - 8, frames=frames, frame_patch=16)          # K != 3*P*P
    with pytest.raises(TypeError):
        H.gemm(None, W, 8, 64, 768, frames=frames.to(torch.uint8), frame_patch=16)          # uint8 without the normalisation constants
