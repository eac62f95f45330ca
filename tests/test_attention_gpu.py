"""GPU: fused proxy / causal attention kernels vs the oracle's attention cores (fp64 on the kernel's own
bf16 inputs), forward and backward, across the (M,N,L) shapes SURVEY.md §8c lists, ragged padding masks,
the degenerate all-padded row, and a forced online-softmax rescale (large score spike in a late key tile)."""
import pytest
import torch

from oracle import clipvip_oracle as O
from tests.gpu_util import report

pytestmark = pytest.mark.gpu


def _split(qkv, B, S, H):
    q, k, v = qkv.view(B, S, 3, H, 64).double().unbind(2)          # [B,S,H,64]
    return [t.transpose(1, 2) for t in (q, k, v)]                  # [B,H,S,64]


def _run(B, H, size, S, pad_mask, seed, scale=1.0, spike=False, q_scale=1.0, dtype=torch.bfloat16):
    from xpretrain_amd import hip_ops as Hh
    torch.manual_seed(seed)
    tf, tb = (1.2e-2, 2e-2) if dtype == torch.bfloat16 else (2e-5, 1e-4)     # fp32 mode: exact-arithmetic kernels
    qkv = (torch.randn(B * S, 3 * H * 64, device="cuda") * scale).to(dtype)
    if spike:   # one key late in the sequence dominates one query's row: forces m to jump at the last tile
        v = qkv.view(B, S, 3, H, 64)
        v[0, S - 1, 1, 0] = v[0, S // 2, 0, 0] * 6.0
    out, stats = Hh.attn_fwd(qkv, B, S, H, size=size, pad_mask=pad_mask)
    q, k, v = [t.requires_grad_() for t in _split(qkv, B, S, H)]
    if size is not None:
        ref = O.proxy_attention_core(q, k, v, size)
    else:
        ref = O.masked_attention_core(q, k, v, None if pad_mask is None else pad_mask.cpu().to(q.device))
    refo = ref.transpose(1, 2).reshape(B * S, H * 64)
    tag = f"attn {str(dtype)[6:]} B{B} H{H} size{size} S{S} pad{pad_mask is not None}"
    e1 = report(tag + " fwd", out, refo, tf)
    dout = torch.randn(B * S, H * 64, device="cuda").to(dtype)
    refo.backward(dout.double())
    dqkv = Hh.attn_bwd(qkv, out, dout, stats, B, S, H, size=size, pad_mask=pad_mask, q_scale=q_scale)
    dq, dk, dv = [t.transpose(1, 2) for t in dqkv.view(B, S, 3, H, 64).double().unbind(2)]
    e2 = report(tag + " dq", dq, q.grad * q_scale, tb)
    e3 = report(tag + " dk", dk, k.grad, tb)
    e4 = report(tag + " dv", dv, v.grad, tb)
    assert e1 <= tf and e2 <= tb and e3 <= tb and e4 <= tb, tag
    assert torch.isfinite(dqkv.float()).all()


@pytest.mark.parametrize("size,B,H", [((4, 2, 49), 2, 2), ((4, 12, 196), 1, 2), ((1, 3, 5), 2, 1), ((4, 3, 70), 1, 3),
                                      ((2, 5, 16), 2, 3), ((4, 1, 196), 1, 1), ((4, 2, 784), 1, 1), ((4, 32, 196), 1, 1),
                                      ((20, 3, 49), 1, 2), ((17, 2, 180), 1, 1),       # > 16 proxy tokens: outside the persistent kernel's mask
                                      ((4, 3, 300), 2, 2), ((2, 2, 500), 1, 3), ((4, 5, 208), 1, 2), ((1, 2, 1023), 1, 1),
                                      ((4, 9, 784), 2, 3)])      # multi-group persistent kernel: 2..5 key groups, 2..4 query blocks, many problems

def test_proxy_attention(size, B, H):
    M, N, L = size
    _run(B, H, size, M + N * L, None, seed=M + N + L)


def test_proxy_attention_rescale_and_qscale():
    _run(1, 2, (4, 3, 196), 4 + 3 * 196, None, seed=5, scale=2.0, spike=True, q_scale=0.125)


@pytest.mark.parametrize("B,S,H,mode", [(3, 12, 2, "ragged"), (2, 32, 8, "ragged"), (2, 7, 1, "none"), (2, 77, 2, "ragged"),
                                        (2, 16, 2, "allpad"), (1, 130, 1, "ragged")])
def test_causal_attention(B, S, H, mode):
    torch.manual_seed(S)
    if mode == "none":
        mask = None
    else:
        lens = torch.randint(1, S + 1, (B,)); lens[0] = S
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        if mode == "allpad":
            mask[1] = 0
        mask = mask.cuda()
    _run(B, H, None, S, mask, seed=S + 1)


@pytest.mark.parametrize("size,B,H", [((4, 2, 49), 2, 2), ((4, 12, 196), 1, 2), ((1, 3, 5), 2, 1), ((4, 2, 784), 1, 1)])
def test_proxy_attention_fp32_mode(size, B, H):
    """the fp32 compute mode's attention kernels (csrc/attention_f32.hip) against the fp64 oracle core"""
    M, N, L = size
    _run(B, H, size, M + N * L, None, seed=M + N + L, dtype=torch.float32, q_scale=0.125)


@pytest.mark.parametrize("B,S,H,mode", [(3, 12, 2, "ragged"), (2, 77, 2, "ragged"), (2, 16, 2, "allpad"), (1, 130, 1, "none")])
def test_causal_attention_fp32_mode(B, S, H, mode):
    torch.manual_seed(S)
    mask = None
    if mode != "none":
        lens = torch.randint(1, S + 1, (B,)); lens[0] = S
        mask = (torch.arange(S)[None] < lens[:, None]).long()
        if mode == "allpad":
            mask[1] = 0
        mask = mask.cuda()
    _run(B, H, None, S, mask, seed=S + 1, dtype=torch.float32)


def test_attention_rejects_bad_shapes():
    from xpretrain_amd import hip_ops as Hh
    qkv = torch.zeros(10 * 3, 3 * 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="M\\+N\\*L"):
        Hh.attn_fwd(qkv, 3, 10, 1, size=(4, 2, 2))


@pytest.mark.parametrize("geom", [(4, 12, 196), (4, 2, 49), (1, 3, 5), None])
def test_backward_emits_the_qkv_bias_column_sums(geom):
    """xp_attn_bwd2: the bias gradients of q/k/v_proj (column sums of dqkv as stored) come out of the backward kernels -- frame rows
    from the dQ / dKV workgroups, proxy rows from the proxy reduce -- and equal a separate pass over dqkv; the causal text pattern
    (ragged padding) too."""
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(11)
    B, Hh = 2, 3
    if geom is None:
        S, size = 32, None
        pad = torch.ones(B, S, dtype=torch.int64, device="cuda"); pad[0, 20:] = 0
    else:
        M, N, Lp = geom
        S, size, pad = M + N * Lp, geom, None
    qkv = (torch.randn(B * S, 3 * Hh * 64, device="cuda") * 0.7).to(torch.bfloat16)
    out, stats = H.attn_fwd(qkv, B, S, Hh, size=size, pad_mask=pad)
    dout = torch.randn_like(out)
    ref = H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=size, pad_mask=pad, q_scale=0.125)
    d = H.DeferredReduce(qkv.device)
    dqkv, cs = H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=size, pad_mask=pad, q_scale=0.125, colsum_defer=d)
    assert len(d.segs) == 1                      # the fused path: one partial-row segment, no extra pass
    d.flush()
    assert torch.equal(dqkv, ref)
    want = dqkv.double().sum(0)
    assert report(f"attn bwd fused colsum {geom}", cs, want, 1e-5, scale_floor=1e-3) <= 1e-5


@pytest.mark.parametrize("geom,B,Hh", [((4, 12, 196), 2, 3), ((4, 2, 49), 2, 2), ((1, 3, 5), 2, 1), ((16, 3, 192), 1, 2), ((4, 7, 204), 1, 2),
                                       ((3, 40, 100), 3, 5)])
def test_fused_backward_against_the_two_kernel_path(geom, B, Hh, monkeypatch):
    """attn_bwd5_kernel (one launch: dQ, dK, dV, the bias column sums; problems handed out by a device counter) against the dQ / dKV
    kernel pair on identical inputs (XPRETRAIN_DEBUG=attn_bwd_split): same operands, same rounding points, different summation order
    -- within 1e-2 of the tensor scale on bf16 outputs (one bf16 ulp is 4e-3); the last case has more problems than CUs (every
    workgroup pulls several from the counter) and a 13-tile-free shape; run twice: the counter decides only WHO computes a problem."""
    from xpretrain_amd import hip_ops as H
    M, N, Lp = geom
    S = M + N * Lp
    torch.manual_seed(3)
    qkv = (torch.randn(B * S, 3 * Hh * 64, device="cuda") * 0.7).to(torch.bfloat16)
    out, stats = H.attn_fwd(qkv, B, S, Hh, size=geom)
    dout = torch.randn_like(out)
    d = H.DeferredReduce(qkv.device)
    got, cs = H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=geom, q_scale=0.125, colsum_defer=d)
    d.flush()
    again = H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=geom, q_scale=0.125)
    assert torch.equal(got, again)
    monkeypatch.setenv("XPRETRAIN_DEBUG", "attn_bwd_split")
    want = H.attn_bwd(qkv, out, dout, stats, B, S, Hh, size=geom, q_scale=0.125)
    monkeypatch.delenv("XPRETRAIN_DEBUG")
    assert torch.isfinite(got.float()).all()
    for j, name in enumerate("qkv"):
        a, b = [t.view(B * S, 3, Hh * 64)[:, j] for t in (got, want)]
        assert report(f"attn bwd fused vs split {geom} d{name}", a, b, 1e-2) <= 1e-2
    assert report(f"attn bwd fused colsum vs stored {geom}", cs, got.double().sum(0), 1e-5, scale_floor=1e-3) <= 1e-5
