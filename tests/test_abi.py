"""CPU: the C-ABI library builds/loads and exports every symbol include/xpretrain_hip.h declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "xpretrain_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(xp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from xpretrain_amd import _lib
    assert _declared() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from xpretrain_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.lib()          # raises if any symbol is missing
    assert lib.xp_abi_version() == 1
    assert lib.xp_last_error() is not None


def test_cpu_tensors_are_rejected():
    import torch
    from xpretrain_amd import hip_ops as H
    with pytest.raises(RuntimeError, match="no CPU path"):
        H.gemm(torch.zeros(8, 8), torch.zeros(8, 8), 8, 8, 8)
