"""GPU: MFMA GEMM family vs torch fp64 matmul on bf16-rounded / fp32 inputs (the kernel's own inputs),
every operand layout, every epilogue, ragged M/N/K tails, split-K."""
import pytest
import torch

from tests.gpu_util import report

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float32]
# error of the OUTPUT rounding dominates for bf16 (2^-9 relative); accumulation is fp32
TOL = {torch.bfloat16: 6e-3, torch.float32: 2e-5}


def _mk(shape, dtype, scale=1.0):
    return (torch.randn(shape, device="cuda") * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (204, 132, 200), (8, 512, 768), (1000, 2304, 768)])
def test_nt_plain(dtype, M, N, K):
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(M + N + K)
    A, B = _mk((M, K), dtype), _mk((N, K), dtype)
    C = H.gemm(A, B, M, N, K)
    ref = A.double() @ B.double().t()
    assert report(f"gemm_nt {dtype} {M}x{N}x{K}", C, ref, TOL[dtype]) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (204, 768, 2304), (52, 192, 136)])
def test_nn_b_kstrided(dtype, M, N, K):
    """dX = dY[M,K] . W[K,N]  (B stored [k][n])"""
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(1)
    A, W = _mk((M, K), dtype), _mk((K, N), dtype)
    C = H.gemm(A, W, M, N, K, b_kstrided=True)
    ref = A.double() @ W.double()
    assert report(f"gemm_nn {dtype} {M}x{N}x{K}", C, ref, TOL[dtype]) <= TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,split", [(128, 128, 256, 1), (768, 3072, 1000, 1), (192, 136, 204, 1), (768, 768, 4000, 8)])
def test_tn_both_kstrided(dtype, M, N, K, split):
    """dW[M,N] = dY[K,M]^T . X[K,N]  (both stored [k][row]), optionally split-K over the long contraction."""
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(2)
    Y, X = _mk((K, M), dtype), _mk((K, N), dtype)
    if split == 1:
        C = H.gemm(Y, X, M, N, K, a_kstrided=True, b_kstrided=True, out_dtype=torch.float32)
    else:
        slabs = H.gemm(Y, X, M, N, K, a_kstrided=True, b_kstrided=True, split_k=split)
        C = H.splitk_reduce(slabs, torch.empty(M, N, device="cuda"))
    ref = Y.double().t() @ X.double()
    tol = 2e-5 if dtype == torch.float32 else 1e-5   # fp32 output: only accumulation error
    assert report(f"gemm_tn {dtype} {M}x{N}x{K} split{split}", C, ref, tol) <= tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_epilogues(dtype):
    from xpretrain_amd import hip_ops as H
    from xpretrain_amd import _lib as L
    torch.manual_seed(3)
    M, N, K = 204, 384, 192
    A, B = _mk((M, K), dtype, 0.5), _mk((N, K), dtype, 0.2)
    bias = torch.randn(N, device="cuda")
    R = _mk((M, N), dtype)
    acc = A.double() @ B.double().t()
    tol = TOL[dtype]
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS, bias=bias)
    assert report(f"epi_bias {dtype}", C, acc + bias.double(), tol) <= tol
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_QSCALE, bias=bias, scale=0.125, scale_cols=128)
    ref = acc + bias.double(); ref[:, :128] *= 0.125
    assert report(f"epi_qscale {dtype}", C, ref, tol) <= tol
    aux = torch.empty(M, N, dtype=dtype, device="cuda")
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = acc + bias.double()
    assert report(f"epi_gelu.aux {dtype}", aux, pre, tol) <= tol
    assert report(f"epi_gelu.act {dtype}", C, pre * torch.sigmoid(1.702 * pre), tol) <= tol
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_RESID, bias=bias, resid=R)
    assert report(f"epi_resid {dtype}", C, acc + bias.double() + R.double(), tol) <= tol
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_GELU_BWD, resid=R)
    x = R.double(); s = torch.sigmoid(1.702 * x)
    assert report(f"epi_gelu_bwd {dtype}", C, acc * (s * (1 + 1.702 * x * (1 - s))), tol) <= tol
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_SCALE, scale=3.0, out_dtype=torch.float32)
    assert report(f"epi_scale {dtype}", C, acc * 3.0, 2e-5) <= 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_epilogue_and_row_remap(dtype):
    """conv-as-GEMM epilogue: rows (b,t,l) land in token slot b*S + Mp + t*L + l with + time[t] + pos[l]."""
    from xpretrain_amd import hip_ops as H
    from xpretrain_amd import _lib as L
    torch.manual_seed(4)
    Bsz, T, Lp, Mp, D, K = 2, 3, 10, 4, 128, 192
    S = Mp + T * Lp
    A, W = _mk((Bsz * T * Lp, K), dtype, 0.3), _mk((D, K), dtype, 0.3)
    tab_t, tab_l = torch.randn(T, D, device="cuda"), torch.randn(Lp, D, device="cuda")
    x = torch.zeros(Bsz * S, D, dtype=dtype, device="cuda")
    H.gemm(A, W, Bsz * T * Lp, D, K, out=x, epilogue=L.EPI_PATCH, tab1=tab_t, tab2=tab_l, tab_L=Lp,
           c_remap=(T * Lp, S, Mp))
    ref = (A.double() @ W.double().t()).view(Bsz, T, Lp, D) + tab_t.double()[None, :, None] + tab_l.double()[None, None]
    full = torch.zeros(Bsz, S, D, dtype=torch.float64, device="cuda")
    full[:, Mp:] = ref.view(Bsz, T * Lp, D)
    assert report(f"epi_patch {dtype}", x.view(Bsz, S, D), full, TOL[dtype]) <= TOL[dtype]
    # dW with the A operand's k-rows remapped the same way (token rows -> patch rows)
    dx = _mk((Bsz * S, D), dtype)
    dW = H.gemm(dx, A, D, K, Bsz * T * Lp, a_kstrided=True, b_kstrided=True, lda=D, ldb=K, out_dtype=torch.float32,
                a_remap=(T * Lp, S, Mp))
    ref = dx.double().view(Bsz, S, D)[:, Mp:].reshape(-1, D).t() @ A.double()
    assert report(f"dW_patch_remap {dtype}", dW, ref, 2e-5) <= 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
def test_colsum(dtype):
    from xpretrain_amd import hip_ops as H
    X = _mk((1000, 768), dtype)
    out = H.colsum(X, 1000, 768)
    assert report(f"colsum {dtype}", out, X.double().sum(0), 1e-5) <= 1e-5


def test_bad_arguments_raise():
    from xpretrain_amd import hip_ops as H
    A = torch.zeros(8, 12, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError, match="multiples of"):
        H.gemm(A, A, 8, 8, 12)


# ---------------------------------------------------------------------------------------------- 256x256 family
@pytest.fixture
def force_gemm256(monkeypatch):
    monkeypatch.setenv("XPRETRAIN_GEMM256", "2")     # use the 256x256 family whenever its preconditions hold
    monkeypatch.setenv("XPRETRAIN_GEMM256_SPLITK", "1")


# tile_rows_hint 0: the library default for training-shaped calls (staged-epilogue kernels, gemm256s.hip); 256: the direct-epilogue
# kernels of gemm256.hip at the same tile height (224: those kernels with 224-row tiles where the cost model picks them, below)
HINTS = [0, 256]


@pytest.mark.parametrize("hint", HINTS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1280, 512, 768), (1100, 768, 256), (300, 260, 96), (2356, 768, 3072)])
def test_gemm256_nt(force_gemm256, dtype, M, N, K, hint):
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(M + N + K)
    A, B = _mk((M, K), dtype), _mk((N, K), dtype)
    C = H.gemm(A, B, M, N, K, tile_rows_hint=hint)
    ref = A.double() @ B.double().t()
    assert report(f"gemm256_nt {dtype} {M}x{N}x{K}", C, ref, TOL[dtype]) <= TOL[dtype]


@pytest.mark.parametrize("hint", HINTS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(1280, 768, 2304), (520, 256, 72), (2356, 3072, 768)])
def test_gemm256_nn(force_gemm256, dtype, M, N, K, hint):
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(1)
    A, W = _mk((M, K), dtype), _mk((K, N), dtype)
    C = H.gemm(A, W, M, N, K, b_kstrided=True, tile_rows_hint=hint)
    ref = A.double() @ W.double()
    assert report(f"gemm256_nn {dtype} {M}x{N}x{K}", C, ref, TOL[dtype]) <= TOL[dtype]


@pytest.mark.parametrize("hint", HINTS)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,split", [(768, 1024, 1000, 1), (256, 512, 77, 1), (768, 768, 4000, 8)])
def test_gemm256_tn(force_gemm256, dtype, M, N, K, split, hint):
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(2)
    Y, X = _mk((K, M), dtype), _mk((K, N), dtype)
    if split == 1:
        C = H.gemm(Y, X, M, N, K, a_kstrided=True, b_kstrided=True, out_dtype=torch.float32, tile_rows_hint=hint)
    else:
        C = H.splitk_reduce(H.gemm(Y, X, M, N, K, a_kstrided=True, b_kstrided=True, split_k=split, tile_rows_hint=hint),
                            torch.empty(M, N, device="cuda"))
    ref = Y.double().t() @ X.double()
    assert report(f"gemm256_tn {dtype} {M}x{N}x{K} split{split}", C, ref, 2e-5) <= 2e-5


@pytest.mark.parametrize("hint", HINTS)
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm256_epilogues(force_gemm256, dtype, hint):
    import functools
    from xpretrain_amd import hip_ops as H0
    from xpretrain_amd import _lib as L

    class H:                                          # every call of this test with the tile hint under test
        gemm = staticmethod(functools.partial(H0.gemm, tile_rows_hint=hint))
    torch.manual_seed(3)
    M, N, K = 1100, 512, 192
    A, B = _mk((M, K), dtype, 0.5), _mk((N, K), dtype, 0.2)
    bias = torch.randn(N, device="cuda")
    R = _mk((M, N), dtype)
    acc = A.double() @ B.double().t()
    tol = TOL[dtype]
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_QSCALE, bias=bias, scale=0.125, scale_cols=128)
    ref = acc + bias.double(); ref[:, :128] *= 0.125
    assert report(f"g256 epi_qscale {dtype}", C, ref, tol) <= tol
    aux = torch.empty(M, N, dtype=dtype, device="cuda")
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = acc + bias.double()
    assert report(f"g256 epi_gelu.aux {dtype}", aux, pre, tol) <= tol
    assert report(f"g256 epi_gelu.act {dtype}", C, pre * torch.sigmoid(1.702 * pre), tol) <= tol
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_RESID, bias=bias, resid=R)
    assert report(f"g256 epi_resid {dtype}", C, acc + bias.double() + R.double(), tol) <= tol
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_GELU_BWD, resid=R)
    x = R.double(); s = torch.sigmoid(1.702 * x)
    assert report(f"g256 epi_gelu_bwd {dtype}", C, acc * (s * (1 + 1.702 * x * (1 - s))), tol) <= tol


@pytest.mark.parametrize("hint", HINTS)
@pytest.mark.parametrize("epi", ["none", "gelu_bwd"])
def test_gemm_fused_colsum(epi, hint):
    """Column sums of the finished outputs from the GEMM epilogue (the bias gradient of the producing Linear), M not a
    multiple of the tile (rows >= M must not contribute), vs fp64."""
    from xpretrain_amd import hip_ops as H
    from xpretrain_amd import _lib as L
    torch.manual_seed(12)
    M, N, K = 8 * 2356 + 40, 768, 512
    bf = torch.bfloat16
    dY = (torch.randn(M, K, device="cuda") * 0.5).to(bf)
    W = (torch.randn(K, N, device="cuda") * 0.05).to(bf)          # [K, N]: read k-strided (the dX orientation)
    pre = torch.randn(M, N, device="cuda").to(bf)
    defer = H.DeferredReduce(dY.device)
    kw = dict(epilogue=L.EPI_GELU_BWD, resid=pre) if epi == "gelu_bwd" else {}
    kw["tile_rows_hint"] = hint
    out, cs = H.gemm(dY, W, M, N, K, b_kstrided=True, colsum_defer=defer, **kw)
    assert len(defer.segs) == 1 and defer.segs[0].nrows == 2 * ((M + 255) // 256)      # the fused path was taken (256-row tiles)
    defer.flush()
    ref = dY.double() @ W.double()
    if epi == "gelu_bwd":
        s = torch.sigmoid(1.702 * pre.double())
        ref = ref * (s * (1 + 1.702 * pre.double() * (1 - s)))
    assert report(f"fused colsum {epi} out", out, ref, 6e-3) <= 6e-3
    assert report(f"fused colsum {epi}", cs, ref.sum(0), 2e-3) <= 2e-3
    # small problem: the library declines the fusion, the wrapper falls back to a separate pass with the same result
    out2, cs2 = H.gemm(dY[:300].contiguous(), W, 300, N, K, b_kstrided=True, colsum_defer=defer, tile_rows_hint=hint,
                       **({} if epi == "none" else dict(epilogue=L.EPI_GELU_BWD, resid=pre[:300].contiguous())))
    defer.flush()
    assert report(f"fallback colsum {epi}", cs2, out2.double().sum(0), 1e-5) <= 1e-5


# ---------------------------------------------------------------------------------------------- 224-row tiles, direct epilogue
def _tile_rows(M, N, K, b_kstrided=False, split_k=1, out_f32=False, hint=224):
    import ctypes as C
    from xpretrain_amd import _lib as L
    d = L.XpGemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldr, d.ldaux = M, N, K, K, (N if b_kstrided else K), N, N, N
    d.b_kstrided, d.in_dtype, d.out_dtype, d.split_k = int(b_kstrided), L.XP_BF16, (L.XP_F32 if out_f32 else L.XP_BF16), split_k
    d.tile_rows_hint = hint
    return int(L.lib().xp_gemm_tile_rows(C.byref(d)))


def test_tile_height_is_chosen_per_shape():
    """BASELINE cfg #2 token count: 85 tiles of 224 rows fill 1 / 3 / 4 rounds of the CUs (74 of 256 rows: 0.87 of them) -- used when
    the caller asks for them (tile_rows_hint = 224: latency-first forward passes), the default stays 256 (energy per training step);
    a multiple of 256 keeps 256-row tiles; small problems stay in the 128x128 family."""
    assert _tile_rows(18848, 768, 768, hint=0) == 256 and _tile_rows(18848, 3072, 768, b_kstrided=True, hint=0) == 256
    assert _tile_rows(18848, 768, 768) == 224 and _tile_rows(18848, 2304, 768) == 224 and _tile_rows(18848, 3072, 768) == 224
    assert _tile_rows(18848, 768, 3072, b_kstrided=True) == 224
    assert _tile_rows(50208, 768, 768) == 224              # configs[3]/[4] token count
    assert _tile_rows(16384, 1024, 512) == 256
    assert _tile_rows(256, 512, 768) == 128


@pytest.mark.parametrize("M", [18848, 18848 + 40, 18848 - 200])
def test_gemm224_nt_all_epilogues(M):
    """224-row tiles (second M half = 48 rows per wave, zero-filled to 64 in LDS) through every fused epilogue of the direct
    (accumulator -> global, N-side rows permuted) store path; ragged last tile."""
    from xpretrain_amd import hip_ops as H
    from xpretrain_amd import _lib as L
    torch.manual_seed(M)
    bf = torch.bfloat16
    N, K = 768, 192
    assert _tile_rows(M, N, K) == 224
    A, B = _mk((M, K), bf, 0.5), _mk((N, K), bf, 0.2)
    bias = torch.randn(N, device="cuda")
    R = _mk((M, N), bf)
    acc = A.double() @ B.double().t()
    tol = TOL[bf]
    C = H.gemm(A, B, M, N, K, tile_rows_hint=224)
    assert report("g224 none", C, acc, tol) <= tol
    C = H.gemm(A, B, M, N, K, tile_rows_hint=224, epilogue=L.EPI_BIAS, bias=bias)
    assert report("g224 bias", C, acc + bias.double(), tol) <= tol
    C = H.gemm(A, B, M, N, K, tile_rows_hint=224, epilogue=L.EPI_BIAS_QSCALE, bias=bias, scale=0.125, scale_cols=256)
    ref = acc + bias.double(); ref[:, :256] *= 0.125
    assert report("g224 qscale", C, ref, tol) <= tol
    aux = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
    C = H.gemm(A, B, M, N, K, tile_rows_hint=224, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = acc + bias.double()
    assert report("g224 gelu.aux", aux, pre, tol) <= tol
    assert report("g224 gelu.act", C, pre * torch.sigmoid(1.702 * pre), tol) <= tol
    C2 = H.gemm(A, B, M, N, K, tile_rows_hint=224, epilogue=L.EPI_BIAS_GELU, bias=bias)                 # forward-only: no pre-activation
    assert torch.equal(C2, C)
    C = H.gemm(A, B, M, N, K, tile_rows_hint=224, epilogue=L.EPI_BIAS_RESID, bias=bias, resid=R)
    assert report("g224 resid", C, acc + bias.double() + R.double(), tol) <= tol
    C = H.gemm(A, B, M, N, K, tile_rows_hint=224, epilogue=L.EPI_GELU_BWD, resid=R)
    x = R.double(); s = torch.sigmoid(1.702 * x)
    assert report("g224 gelu_bwd", C, acc * (s * (1 + 1.702 * x * (1 - s))), tol) <= tol
    # rows past M are never written: a guard band behind the output stays untouched
    big = torch.full((M + 300, N), 7.0, dtype=bf, device="cuda")
    H.gemm(A, B, M, N, K, out=big, tile_rows_hint=224)
    assert bool((big[M:] == 7.0).all())


def test_gemm224_nn_and_f32_slabs():
    """the dX orientation (N side k-strided: transpose reads gather the permuted columns) on 224-row tiles, and fp32 output
    through the direct path (two 16-byte stores per lane)"""
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(5)
    bf = torch.bfloat16
    M, N, K = 18848, 768, 256
    assert _tile_rows(M, N, K, b_kstrided=True) == 224
    A, W = _mk((M, K), bf, 0.5), _mk((K, N), bf, 0.2)
    C = H.gemm(A, W, M, N, K, b_kstrided=True, tile_rows_hint=224)
    ref = A.double() @ W.double()
    assert report("g224 nn", C, ref, TOL[bf]) <= TOL[bf]
    Cf = H.gemm(A, W, M, N, K, b_kstrided=True, out_dtype=torch.float32, tile_rows_hint=224)
    assert report("g224 nn f32", Cf, ref, 2e-5) <= 2e-5
    B = _mk((N, K), bf, 0.2)
    Cf = H.gemm(A, B, M, N, K, out_dtype=torch.float32, tile_rows_hint=224)
    assert report("g224 nt f32", Cf, A.double() @ B.double().t(), 2e-5) <= 2e-5


@pytest.mark.parametrize("M,N,K", [(18848, 1024, 192), (18848, 3072, 128), (16384, 1280, 256), (40000, 768, 128)])
def test_gemm256_persistent_tile_loop(M, N, K):
    """More tiles than CUs in the forward orientation: one workgroup per CU walks several tiles, the next tile's first half-tiles
    are prefetched under the current tile's stores (gemm256_persist_kernel).  Every epilogue it serves, both tile heights, a ragged
    last tile, and bit-equality with the one-tile-per-workgroup kernel (same arithmetic, different schedule)."""
    import functools
    import os
    import subprocess
    import sys
    from xpretrain_amd import hip_ops as H0
    from xpretrain_amd import _lib as L

    class H:                                          # the direct-epilogue kernels (the persistent loop is theirs) at 256-row tiles
        gemm = staticmethod(functools.partial(H0.gemm, tile_rows_hint=256))
    torch.manual_seed(M + N)
    bf = torch.bfloat16
    A, B = _mk((M, K), bf, 0.5), _mk((N, K), bf, 0.2)
    bias = torch.randn(N, device="cuda")
    acc = A.double() @ B.double().t()
    tol = TOL[bf]
    C0 = H.gemm(A, B, M, N, K)
    assert report("persist none", C0, acc, tol) <= tol
    assert torch.equal(H0.gemm(A, B, M, N, K, tile_rows_hint=224), C0)      # 224-row tiles: same k order, bit-identical
    assert report("staged kernels vs persistent", H0.gemm(A, B, M, N, K), C0, 1e-6) <= 1e-6     # (default family: same k order too)
    C1 = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS, bias=bias)
    assert report("persist bias", C1, acc + bias.double(), tol) <= tol
    C2 = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_QSCALE, bias=bias, scale=0.125, scale_cols=256)
    ref = acc + bias.double(); ref[:, :256] *= 0.125
    assert report("persist qscale", C2, ref, tol) <= tol
    aux = torch.full((M, N), float("nan"), dtype=bf, device="cuda")
    C3 = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_GELU, bias=bias, aux=aux)
    pre = acc + bias.double()
    assert report("persist gelu.aux", aux, pre, tol) <= tol
    assert report("persist gelu.act", C3, pre * torch.sigmoid(1.702 * pre), tol) <= tol
    C4 = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_GELU, bias=bias)
    assert torch.equal(C4, C3)
    for _ in range(5):                                   # repeated launches: the prefetch / store overlap has no race
        assert torch.equal(H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS, bias=bias), C1)
    # the same problems through the one-tile-per-workgroup kernel in a fresh process (the switch is read once per process)
    torch.save({"A": A.cpu(), "B": B.cpu(), "bias": bias.cpu(), "C1": C1.cpu(), "C3": C3.cpu()}, "/tmp/xp_persist_case.pt")
    code = ("import torch, sys; sys.path.insert(0, '.');\n"
            "from xpretrain_amd import hip_ops as H, _lib as L\n"
            "d = torch.load('/tmp/xp_persist_case.pt'); A, B, bias = d['A'].cuda(), d['B'].cuda(), d['bias'].cuda()\n"
            "M, K = A.shape; N = B.shape[0]\n"
            "C1 = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS, bias=bias, tile_rows_hint=256)\n"
            "C3 = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_GELU, bias=bias, tile_rows_hint=256)\n"
            "assert torch.equal(C1.cpu(), d['C1']) and torch.equal(C3.cpu(), d['C3'])\nprint('same')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, XPRETRAIN_GEMM256_PERSIST="0"),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "same" in out.stdout, out.stderr[-1500:]


@pytest.mark.parametrize("M,N,K,S,Ms,hint", [(18848, 768, 192, 2356, 4, 0), (18848, 768, 192, 2356, 4, 224), (18848, 768, 192, 2356, 4, 256),
                                             (4712, 768, 3072, 2356, 4, 0), (4712, 768, 3072, 2356, 4, 256),
                                             (200, 256, 128, 50, 3, 0), (40, 64, 64, 10, 4, 0)])
def test_resid_epilogue_with_fp32_side_rows(M, N, K, S, Ms, hint):
    """EPI_BIAS_RESID with the fp32 side rows of the residual stream (XpGemmDesc::resid_side / out_side): rows m with m % S < Ms take
    their residual operand from the fp32 side buffer and leave their fp32 result there as well as the rounded C row; all other rows
    are untouched by the feature.  256-wide family (both tile heights, K = 768-like and K = 3072) and the 128x128 family."""
    from xpretrain_amd import hip_ops as H
    from xpretrain_amd import _lib as L
    torch.manual_seed(M + N)
    bf = torch.bfloat16
    A, B = _mk((M, K), bf, 0.5), _mk((N, K), bf, 0.2)
    bias = torch.randn(N, device="cuda")
    R = _mk((M, N), bf)
    nb = (M + S - 1) // S
    rs = torch.randn(nb * Ms, N, device="cuda")
    os_ = torch.full((nb * Ms, N), float("nan"), device="cuda")
    C = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_RESID, bias=bias, resid=R, resid_side=rs, out_side=os_, side=(S, Ms), tile_rows_hint=hint)
    plain = H.gemm(A, B, M, N, K, epilogue=L.EPI_BIAS_RESID, bias=bias, resid=R, tile_rows_hint=hint)
    acc = A.double() @ B.double().t() + bias.double()
    rows = torch.arange(M, device="cuda")
    is_side = (rows % S) < Ms
    sidx = (rows // S) * Ms + rows % S
    # side rows: fp32 result = acc + bias + fp32 residual; C = that, rounded
    want = acc[is_side] + rs[sidx[is_side]].double()
    assert report("side rows fp32", os_[sidx[is_side]], want, 2e-5) <= 2e-5
    assert torch.equal(C[is_side], os_[sidx[is_side]].to(bf))
    assert torch.equal(C[~is_side], plain[~is_side])
    # side slots of rows >= M (a partial last sample) stay untouched
    touched = torch.zeros(nb * Ms, dtype=torch.bool, device="cuda"); touched[sidx[is_side]] = True
    assert bool(torch.isnan(os_[~touched]).all())
