"""CPU: the host-side planning entry points of the C ABI (no kernel is launched, no GPU needed): split-K planning,
fused-column-sum availability, partial-row counts and workspace sizes."""
import ctypes as C

import pytest

from xpretrain_amd import _lib as L


def _desc(M, N, K, *, a_ks=True, b_ks=True, dtype=L.XP_BF16, out=L.XP_F32, epi=L.EPI_NONE, split=1):
    d = L.XpGemmDesc()
    d.M, d.N, d.K = M, N, K
    d.lda = M if a_ks else K
    d.ldb = N if b_ks else K
    d.ldc = N
    d.a_kstrided, d.b_kstrided = int(a_ks), int(b_ks)
    d.in_dtype, d.out_dtype, d.epilogue, d.split_k = dtype, out, epi, split
    return d


def _valid_split(K, s, ke=64):
    kps = -(-(-(-K // s)) // ke) * ke
    return -(-K // kps) == s


@pytest.mark.parametrize("M,N,K,expect", [(2304, 768, 18848, 5), (768, 768, 18848, 16), (3072, 768, 18848, 4),
                                          (768, 3072, 18848, 4)])
def test_auto_split_cfg2_weight_gradients(M, N, K, expect):
    """split-K launches fill at most 144 CUs (csrc/gemm.hip::XP_SPLITK_FILL, 176 in round 5: they run beside the dX chain, and every
    split is an fp32 slab written and read again): 36 / 27 / 9 tiles x 4 / 5 / 16"""
    s = L.lib().xp_gemm_auto_split(C.byref(_desc(M, N, K)))
    assert s == expect and _valid_split(K, s) and s * (-(-M // 256)) * (-(-N // 256)) <= 144


@pytest.mark.parametrize("M,N,K,expect", [(768, 768, 18848, 12), (3072, 768, 18848, 3), (768, 3072, 18848, 3), (2304, 768, 18848, 4)])
def test_auto_split_of_launches_with_slack(M, N, K, expect):
    """xp_gemm_auto_split_slack: the first three weight-gradient GEMMs of a layer's backward (nothing waits for them soon) fill at most
    112 CUs: fc2 / fc1 3 slabs, out_proj 12; never more slabs than the general plan"""
    lib = L.lib()
    s = lib.xp_gemm_auto_split_slack(C.byref(_desc(M, N, K)))
    assert s == expect and _valid_split(K, s) and s <= lib.xp_gemm_auto_split(C.byref(_desc(M, N, K)))


def test_auto_split_is_always_accepted():
    lib = L.lib()
    for M, N in [(512, 512), (1536, 512), (2048, 512), (512, 2048), (768, 768), (256, 256), (3072, 768), (128, 96)]:
        for K in [31, 64, 200, 256, 1000, 4096, 6276 * 8, 18848, 100000]:
            for dtype in (L.XP_BF16, L.XP_F32):
                s = lib.xp_gemm_auto_split(C.byref(_desc(M, N, K, dtype=dtype)))
                assert s >= 1
                ke = 64 if dtype == L.XP_BF16 else 32
                assert s == 1 or _valid_split(K, s, ke) or _valid_split(K, s, 64), (M, N, K, dtype, s)


def test_fused_colsum_availability():
    lib = L.lib()
    big = dict(a_ks=False, b_ks=True, out=L.XP_BF16)
    assert lib.xp_gemm_colsum_rows(C.byref(_desc(18848, 3072, 768, epi=L.EPI_GELU_BWD, **big))) == 2 * 74      # 74 tiles of 256 rows (default height)
    assert lib.xp_gemm_colsum_rows(C.byref(_desc(18848, 768, 768, epi=L.EPI_NONE, **big))) == 2 * 74
    assert lib.xp_gemm_colsum_rows(C.byref(_desc(256, 2048, 512, epi=L.EPI_GELU_BWD, **big))) == 0          # text tower: 128 family
    assert lib.xp_gemm_colsum_rows(C.byref(_desc(18848, 3072, 768, epi=L.EPI_BIAS, **big))) == 0            # other epilogue
    assert lib.xp_gemm_colsum_rows(C.byref(_desc(18848, 3072, 768, a_ks=False, b_ks=True, out=L.XP_F32))) == 0
    assert lib.xp_gemm_colsum_rows(C.byref(_desc(18848, 3072, 768, split=2, **big))) == 0
    assert lib.xp_gemm_colsum_rows(C.byref(_desc(18848, 3072, 768, dtype=L.XP_F32, **big))) == 0


def test_tile_height_planning():
    """xp_gemm_tile_rows: 256 for everything the 256-wide family serves (video-tower activations and weight gradients), 128 = the
    128x128 family (text tower, fp32 inputs)."""
    lib = L.lib()
    act = dict(a_ks=False, b_ks=False, out=L.XP_BF16)
    for N in (768, 2304, 3072):
        assert lib.xp_gemm_tile_rows(C.byref(_desc(18848, N, 768, **act))) == 256
    assert lib.xp_gemm_tile_rows(C.byref(_desc(18848, 768, 3072, a_ks=False, b_ks=True, out=L.XP_BF16))) == 256
    assert lib.xp_gemm_tile_rows(C.byref(_desc(50208, 768, 768, **act))) == 256
    assert lib.xp_gemm_tile_rows(C.byref(_desc(3072, 768, 18848, split=4))) == 256        # dW1: 36 tiles x 4 slabs
    assert lib.xp_gemm_tile_rows(C.byref(_desc(256, 2048, 512, **act))) == 128            # text tower
    assert lib.xp_gemm_tile_rows(C.byref(_desc(18848, 768, 768, dtype=L.XP_F32, a_ks=False, b_ks=False))) == 128


def test_cu_budget_shrinks_the_split():
    """xp_set_cu_budget: a data-parallel run reserves the CUs its collective kernels own (distributed.reserve_cus_for_collectives);
    the dW launches must then fit the remaining CUs in one round."""
    lib = L.lib()
    try:
        for budget, expect in ((256, (4, 5, 16)), (224, (4, 5, 16)), (160, (4, 5, 16)), (128, (3, 4, 14))):      # (whole 64-token k-steps per slab)
            assert lib.xp_set_cu_budget(budget) == 0 and lib.xp_get_cu_budget() == budget
            got = tuple(lib.xp_gemm_auto_split(C.byref(_desc(M, N, 18848))) for M, N in ((3072, 768), (2304, 768), (768, 768)))
            tiles = (36, 27, 9)
            assert all(s * t <= budget for s, t in zip(got, tiles)), (budget, got)
            assert got == expect, (budget, got)
        assert lib.xp_set_cu_budget(32) != 0           # out of range: refused
    finally:
        lib.xp_set_cu_budget(256)


def test_partial_row_counts_and_workspaces():
    lib = L.lib()
    for rows in (1, 31, 256, 18848, 50208):
        for cols in (64, 768, 2304, 3072):
            n = lib.xp_colsum_partial_rows(rows, cols)
            assert 1 <= n <= -(-rows // 32) and n >= -(-rows // 128)
            assert lib.xp_colsum_workspace_bytes(rows, cols) >= (n + 32) * cols * 4
        nb = lib.xp_layernorm_bwd_partial_rows(rows)
        assert 1 <= nb <= 512
        assert lib.xp_layernorm_bwd_workspace_bytes(rows, 768) >= nb * 3 * 768 * 4
    segs = (L.XpReduceSeg * 3)()
    for i, w in enumerate((768, 3072, 64)):
        segs[i].width, segs[i].nrows, segs[i].stride = w, 100, w
    assert lib.xp_reduce_rows_batch_workspace_bytes(segs, 3) >= 32 * (768 + 3072 + 64) * 4
    # attention: forward partials / backward (delta + proxy partials) share one workspace
    assert lib.xp_attn_workspace_bytes(L.ATTN_PROXY, 8, 12, 4, 12, 196) >= 8 * 12 * 2356 * 4
    assert lib.xp_attn_workspace_bytes(L.ATTN_CAUSAL, 8, 8, 0, 1, 32) == 8 * 8 * 32 * 4
    assert lib.xp_nce_loss_workspace_bytes(64, 512) >= 2 * 64 * 64 * 4
    assert lib.xp_vsc_fc_loss_workspace_bytes(64, 512) >= 6 * 64 * 64 * 4


def _dims(rows, D, Dff, B, S, heads, size, dtype=L.XP_BF16):
    d = L.XpLayerDims()
    d.rows, d.D, d.Dff, d.B, d.S, d.heads = rows, D, Dff, B, S, heads
    d.M, d.N, d.L = size if size else (0, 1, S)
    d.attn_mode = L.ATTN_PROXY if size else L.ATTN_CAUSAL
    d.dtype, d.q_scale, d.ln_eps = dtype, 0.125, 1e-5
    return d


def test_encoder_layer_workspace_planning():
    """xp_encoder_layer_{fwd,bwd}_workspace_bytes (csrc/layer.hip): the backward workspace holds the six activation-gradient
    temporaries, the largest split-K slab set, the deferred partial rows, the reduce scratch and the attention workspace."""
    lib = L.lib()
    d = _dims(8 * 2356, 768, 3072, 8, 2356, 12, (4, 12, 196))
    fwd, bwd = lib.xp_encoder_layer_fwd_workspace_bytes(C.byref(d)), lib.xp_encoder_layer_bwd_workspace_bytes(C.byref(d))
    rows, D, Dff = 8 * 2356, 768, 3072
    temporaries = rows * (Dff + 4 * D + 3 * D) * 2
    slabs = 4 * 3072 * 768 * 4                                     # fc1 / fc2 dW: split-K 4 (test above)
    assert fwd >= lib.xp_attn_workspace_bytes(L.ATTN_PROXY, 8, 12, 4, 12, 196)
    assert temporaries + slabs < bwd < temporaries + slabs + (64 << 20)
    # text tower shape, fp32 mode: still consistent, and empty dims give 0
    t = _dims(8 * 32, 512, 2048, 8, 32, 8, None, L.XP_F32)
    assert lib.xp_encoder_layer_bwd_workspace_bytes(C.byref(t)) > 8 * 32 * (2048 + 7 * 512) * 4
    z = _dims(0, 512, 2048, 8, 32, 8, None)
    assert lib.xp_encoder_layer_bwd_workspace_bytes(C.byref(z)) == 0


def test_encoder_layer_rejects_bad_arguments():
    lib = L.lib()
    a = L.XpLayerFwd()
    a.dims = _dims(100, 768, 3072, 8, 2356, 12, (4, 12, 196))       # rows != B*S
    assert lib.xp_encoder_layer_fwd(C.byref(a), None) != 0 and b"rows" in lib.xp_last_error()
    b = L.XpLayerBwd()
    b.dims = _dims(8 * 32, 512, 2048, 8, 32, 8, None)
    assert lib.xp_encoder_layer_bwd(C.byref(b), None) != 0 and b"null pointer" in lib.xp_last_error()
