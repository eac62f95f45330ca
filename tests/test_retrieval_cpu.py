"""CPU: pin oracle/retrieval_oracle.py against tests/golden/retrieval.pt (the reference's own metrics functions)."""
import numpy as np
import pytest

from oracle import retrieval_oracle as RO


def test_oracle_metrics_match_reference(golden):
    fx = golden("retrieval.pt")
    for c in fx["cases"]:
        txt, vis = c["txt"].numpy(), c["vis"].numpy()
        got = RO.validate(txt, vis)
        for setting in ("simple", "DSL"):
            for direction in ("v2t", "t2v"):
                assert got[setting][direction] == pytest.approx(c["results"][setting][direction], rel=1e-12), (setting, direction)
        sim = RO.cal_cossim(txt, vis)
        assert RO.summarise(RO.ranks(sim, c["labels"].numpy())) == pytest.approx(c["multi"], rel=1e-12)
        if c["sim"] is not None:
            np.testing.assert_allclose(sim, c["sim"].numpy(), rtol=0, atol=1e-6)
            np.testing.assert_allclose(RO.col_softmax(sim, 100), c["softmax100"].numpy(), rtol=1e-5, atol=1e-12)


def test_oracle_image_norm_matches_reference(golden):
    fx = golden("retrieval.pt")
    got = RO.image_norm(fx["frames"].numpy(), fx["mean"], fx["std"])
    np.testing.assert_allclose(got, fx["normed"].numpy(), rtol=0, atol=2e-7)


def test_metrics_module_has_no_cpu_path():
    import torch
    from xpretrain_amd.utils import metrics
    with pytest.raises(RuntimeError, match="no CPU path"):
        metrics.cal_cossim(torch.zeros(2, 4), torch.zeros(2, 4))
