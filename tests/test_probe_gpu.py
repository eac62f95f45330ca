"""GPU: pin the hardware lane maps libxpretrain_hip.so relies on (common.h) against real gfx950."""
import pytest
import torch

from tests.gpu_util import dump

pytestmark = pytest.mark.gpu


def test_mfma_16x16x32_bf16_layout():
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(0)
    A = torch.randint(-4, 5, (16, 32)).float()
    B = torch.randint(-4, 5, (32, 16)).float()       # asymmetric: catches a transposed C write
    l = torch.arange(64)
    i16, g = l & 15, l >> 4
    k = (8 * g)[:, None] + torch.arange(8)[None]                 # [64,8]
    a = A[i16[:, None], k].to(torch.bfloat16).cuda()             # a[l][e] = A[i16][8g+e]
    b = B[k, i16[:, None]].to(torch.bfloat16).cuda()             # b[l][e] = B[8g+e][i16]
    c = H.probe_mfma_bf16(a, b).cpu()
    D = A @ B
    exp = D[(4 * g)[:, None] + torch.arange(4)[None], i16[:, None]]   # c[l][r] = D[4g+r][i16]
    if not torch.equal(c, exp):
        dump("probe_mfma_bf16.json", dict(A=A.tolist(), B=B.tolist(), c=c.tolist(), D=D.tolist()))
    assert torch.equal(c, exp)


def test_mfma_16x16x4_f32_layout():
    from xpretrain_amd import hip_ops as H
    torch.manual_seed(1)
    A = torch.randint(-4, 5, (16, 4)).float()
    B = torch.randint(-4, 5, (4, 16)).float()
    l = torch.arange(64)
    i16, g = l & 15, l >> 4
    a = A[i16, g].contiguous().cuda()                            # a[l] = A[i16][g]
    b = B[g, i16].contiguous().cuda()
    c = H.probe_mfma_f32(a, b).cpu()
    D = A @ B
    exp = D[(4 * g)[:, None] + torch.arange(4)[None], i16[:, None]]
    if not torch.equal(c, exp):
        dump("probe_mfma_f32.json", dict(A=A.tolist(), B=B.tolist(), c=c.tolist(), D=D.tolist()))
    assert torch.equal(c, exp)


def _tr_model(off):
    """common.h's model: lane i of a 16-lane group gets, for j=0..3, element (i&3) of the 8-byte chunk
    whose address was supplied by lane 4j + (i>>2) of the same group."""
    l = torch.arange(64)
    grp, i = (l >> 4) * 16, l & 15
    out = torch.empty(64, 4, dtype=torch.long)
    for j in range(4):
        src = grp + 4 * j + (i >> 2)
        out[:, j] = off[src] // 2 + (i & 3)
    return out


@pytest.mark.parametrize("pattern", ["linear", "gemm_ks", "random"])
def test_ds_read_tr16_lane_map(pattern):
    from xpretrain_amd import hip_ops as H
    data = torch.arange(4096, dtype=torch.int16).cuda()          # LDS word w holds the value w
    l = torch.arange(64)
    if pattern == "linear":
        off = l * 8
    elif pattern == "gemm_ks":                                   # gemm.hip lfrag<bf16,KS>: ot=3, ks=1
        i, g = l & 15, l >> 4
        f = (i >> 2) | ((g & 1) << 2)
        off = (32 + g * 8 + (i >> 2)) * 256 + ((3 ^ f) << 5) + ((i & 3) << 3)
        off = off % 8192
    else:
        torch.manual_seed(2)
        off = torch.randint(0, 1024, (64,)) * 8
    got = H.probe_tr16(data, off.int().cuda()).cpu().long()
    exp = _tr_model(off)
    if not torch.equal(got, exp):
        dump(f"probe_tr16_{pattern}.json", dict(off=off.tolist(), got=got.tolist(), exp=exp.tolist()))
    assert torch.equal(got, exp)
