"""CPU: pin oracle/optim_oracle.py against tests/golden/optim.pt (the UNMODIFIED reference's AdamW / clip / schedule,
see tests/golden/make_golden.py::optim) and check the host-side mirrors in xpretrain_amd.optimization."""
import pytest
import torch

from oracle import optim_oracle as OO


def _groups(fx):
    lr_mul = fx["lr_mul"]
    return [dict(idx=idx, lr=[(lr_mul if gi < 2 else 1.0) * lr for lr in fx["lrs"]], betas=tuple(fx["betas"]),
                 weight_decay=fx["weight_decay"] if gi % 2 == 0 else 0.0) for gi, idx in enumerate(fx["group_idx"])]


def test_oracle_adamw_matches_reference(golden):
    fx = golden("optim.pt")
    for dtype, tol in ((torch.float64, 2e-6), (torch.float32, 2e-5)):
        ps, norms = OO.train_steps(fx["init"], fx["grads"], _groups(fx), max_norm=5.0, dtype=dtype)
        for n, r in zip(norms, fx["norms"]):
            assert abs(float(n) - float(r)) <= 1e-5 * float(r)
        for i, (p, r) in enumerate(zip(ps, fx["final"])):
            err = (p.float() - r).abs().max().item()
            assert err <= tol * max(r.abs().max().item(), 1e-3), f"param {fx['names'][i]}: {err:.3e}"


def test_oracle_groups_match_reference(golden):
    fx = golden("optim.pt")
    assert [sorted(g) for g in OO.group_names(fx["names"], "text_model")] == [sorted(g) for g in fx["group_idx"]]


def test_oracle_schedules_match_reference(golden):
    fx = golden("optim.pt")
    for decay, ref in fx["sched_table"].items():
        got = [OO.get_lr_sched(t, decay, 3e-4, 50, warmup_ratio=0.1) for t in range(0, 56)]
        assert got == pytest.approx(ref, rel=1e-12, abs=0.0), decay


def test_package_schedules_and_groups(golden):
    """Host logic of the product: same tables / same grouping as the reference."""
    from xpretrain_amd.optimization import sched
    from xpretrain_amd.optimization.utils import build_e2e_optimizer_w_lr_mul
    fx = golden("optim.pt")
    for decay, ref in fx["sched_table"].items():
        got = [sched.get_lr_sched(t, decay, 3e-4, 50, warmup_ratio=0.1) for t in range(0, 56)]
        assert got == pytest.approx(ref, rel=1e-12, abs=0.0), decay
    with pytest.raises(ValueError):
        sched.get_lr_sched(1, "nope", 1e-3, 10)
    params = [torch.nn.Parameter(p.clone()) for p in fx["init"]]
    groups = build_e2e_optimizer_w_lr_mul(list(zip(fx["names"], params)), 1e-3, fx["weight_decay"], lr_mul=fx["lr_mul"],
                                          lr_mul_prefix="text_model")
    ids = [id(p) for p in params]
    assert [[ids.index(id(q)) for q in g["params"]] for g in groups] == fx["group_idx"]
    assert [g["weight_decay"] for g in groups] == [fx["weight_decay"], 0.0, fx["weight_decay"], 0.0]
    assert groups[0]["lr"] == pytest.approx(1e-4) and "lr" not in groups[2]


def test_adamw_argument_checks():
    from xpretrain_amd.optimization import AdamW
    p = torch.nn.Parameter(torch.zeros(3))
    for kw in (dict(lr=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, -0.1)), dict(eps=-1.0)):
        with pytest.raises(ValueError):
            AdamW([p], **kw)
    opt = AdamW([p], lr=1e-3)
    assert opt.defaults["eps"] == 1e-6 and opt.defaults["correct_bias"] is True     # adamw.py:22-23
    assert opt.step() is None                                                        # no gradients: nothing to launch
