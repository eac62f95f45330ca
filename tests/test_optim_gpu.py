"""GPU: the fused clip + AdamW (+ shadow copies) launch against the oracle and the reference's golden run."""
import pytest
import torch

from oracle import optim_oracle as OO
from gpu_util import report

pytestmark = pytest.mark.gpu


def _groups(fx):
    lr_mul = fx["lr_mul"]
    return [dict(idx=idx, lr=[(lr_mul if gi < 2 else 1.0) * lr for lr in fx["lrs"]], betas=tuple(fx["betas"]),
                 weight_decay=fx["weight_decay"] if gi % 2 == 0 else 0.0) for gi, idx in enumerate(fx["group_idx"])]


def test_adamw_golden_run(golden):
    """Same parameters, gradients, schedule and clipping as the reference run in tests/golden/optim.pt."""
    from xpretrain_amd.optimization import AdamW
    from xpretrain_amd.optimization.utils import build_e2e_optimizer_w_lr_mul
    fx = golden("optim.pt")
    params = [torch.nn.Parameter(p.clone().cuda()) for p in fx["init"]]
    groups = build_e2e_optimizer_w_lr_mul(list(zip(fx["names"], params)), 1e-3, fx["weight_decay"], lr_mul=fx["lr_mul"],
                                          lr_mul_prefix="text_model")
    opt = AdamW(groups, lr=1e-3, betas=tuple(fx["betas"]))
    for t, gs in enumerate(fx["grads"]):
        for gi, g in enumerate(opt.param_groups):
            g["lr"] = (fx["lr_mul"] if gi < 2 else 1.0) * fx["lrs"][t]
        for p, g in zip(params, gs):
            p.grad = g.clone().cuda()
        norm = opt.clip_and_step(5.0)
        assert abs(norm.item() - float(fx["norms"][t])) <= 2e-5 * float(fx["norms"][t])
    for i, p in enumerate(params):
        n = fx["names"][i]
        assert report(f"adamw golden {n}", p.detach(), fx["final"][i], 2e-5, scale_floor=1e-3) <= 2e-5
        assert report(f"adamw golden m {n}", opt.state[p]["exp_avg"], fx["exp_avg"][i], 2e-5, scale_floor=1e-6) <= 2e-5
        assert report(f"adamw golden v {n}", opt.state[p]["exp_avg_sq"], fx["exp_avg_sq"][i], 2e-5, scale_floor=1e-9) <= 2e-5
        assert opt.state[p]["step"] == len(fx["grads"])


@pytest.mark.parametrize("clip", [None, 1.0])
def test_adamw_many_tensors_ragged_sizes(clip):
    """> 256 tensors (two launches sharing one partials array), odd sizes, a scalar, a multi-chunk tensor, unaligned
    element counts; two steps; vs the fp64 oracle."""
    from xpretrain_amd.optimization import AdamW
    torch.manual_seed(5)
    shapes = [(), (1,), (3,), (7, 5), (65536 * 2 + 13,), (1000, 130)] + [(17 + i % 9, 3 + i % 5) for i in range(270)]
    init = [torch.randn(s) * 0.1 for s in shapes]
    grads = [[torch.randn(s) * 0.05 for s in shapes] for _ in range(2)]
    half = len(shapes) // 2
    params = [torch.nn.Parameter(p.clone().cuda()) for p in init]
    opt = AdamW([dict(params=params[:half], weight_decay=0.3), dict(params=params[half:], weight_decay=0.0, lr=3e-3)],
                lr=1e-2, betas=(0.8, 0.95), eps=1e-6)
    for gs in grads:
        for p, g in zip(params, gs):
            p.grad = g.clone().cuda()
        opt.step(max_grad_norm=clip)
    ref, norms = OO.train_steps(init, grads, [dict(idx=list(range(half)), lr=1e-2, betas=(0.8, 0.95), weight_decay=0.3),
                                              dict(idx=list(range(half, len(shapes))), lr=3e-3, betas=(0.8, 0.95))],
                                max_norm=clip)
    if clip:
        assert abs(opt.last_grad_norm.item() - float(norms[-1])) <= 1e-5 * float(norms[-1])
        assert float(norms[-1]) > clip      # the clip is active
    worst = max(((p.detach().cpu().double() - r).abs().max() / max(r.abs().max().item(), 1e-3)).item()
                for p, r in zip(params, ref))
    print(f"adamw ragged clip={clip}: worst maxrel {worst:.3e}")
    assert worst <= 2e-5


def test_adamw_rewrites_weight_shadows_and_skips_casts():
    """The update kernel rewrites the bf16 / fused copies held by functional.WEIGHTS; copies of tensors outside the
    optimizer go stale and are re-cast; parameters without gradient are untouched."""
    from xpretrain_amd.optimization import AdamW
    from xpretrain_amd.functional import WEIGHTS
    torch.manual_seed(9)
    wq, wk, wv = (torch.nn.Parameter(torch.randn(64, 32, device="cuda") * 0.1) for _ in range(3))
    bq, bk, bv = (torch.nn.Parameter(torch.randn(64, device="cuda") * 0.1) for _ in range(3))
    w1 = torch.nn.Parameter(torch.randn(96, 64, device="cuda") * 0.1)
    frozen = torch.nn.Parameter(torch.randn(16, 16, device="cuda"))
    other = torch.nn.Parameter(torch.randn(16, 16, device="cuda"))        # stepped by a different optimizer
    bf = torch.bfloat16
    fq, fb, s1 = WEIGHTS.fused((wq, wk, wv), bf), WEIGHTS.fused((bq, bk, bv), torch.float32), WEIGHTS.get(w1, bf)
    sfz, so = WEIGHTS.get(frozen, bf), WEIGHTS.get(other, bf)
    opt = AdamW([wq, wk, wv, bq, bk, bv, w1, frozen], lr=1e-2, weight_decay=0.1)
    sgd = torch.optim.SGD([other], lr=0.1)
    for step in range(2):
        for p in (wq, wk, wv, bq, bk, bv, w1, other):
            p.grad = torch.randn_like(p)
        opt.step(max_grad_norm=1.0)
        sgd.step()
        # the same buffers, already holding the new values, no re-cast needed
        assert WEIGHTS.fused((wq, wk, wv), bf).data_ptr() == fq.data_ptr()
        assert torch.equal(fq, torch.cat([wq, wk, wv]).detach().to(bf))
        assert torch.equal(fb, torch.cat([bq, bk, bv]).detach())
        assert torch.equal(s1, w1.detach().to(bf))
        assert torch.equal(WEIGHTS.get(w1, bf), w1.detach().to(bf))
        assert torch.equal(WEIGHTS.get(other, bf), other.detach().to(bf))     # re-cast after the foreign step
        assert torch.equal(WEIGHTS.get(frozen, bf), frozen.detach().to(bf))
    assert "exp_avg" not in opt.state[frozen]


def test_adamw_state_dict_round_trip():
    """E2E_TrainingRestorer-style resume (utils/load_save.py:306-314): optimizer.state_dict() -> a NEW optimizer ->
    load_state_dict -> the next steps equal the uninterrupted run (the step plan must follow the replaced state tensors)."""
    from xpretrain_amd.optimization import AdamW
    torch.manual_seed(2)
    shapes = [(33, 17), (17,), (), (300, 70)]
    init = [torch.randn(s) * 0.1 for s in shapes]
    grads = [[torch.randn(s) * 0.1 for s in shapes] for _ in range(4)]

    def make():
        ps = [torch.nn.Parameter(p.clone().cuda()) for p in init]
        return ps, AdamW([dict(params=ps[:2], weight_decay=0.1), dict(params=ps[2:], weight_decay=0.0)], lr=1e-2)

    def run(ps, opt, gs):
        for g in gs:
            for p, x in zip(ps, g):
                p.grad = x.clone().cuda()
            opt.clip_and_step(1.0)

    pa, oa = make()
    run(pa, oa, grads)                                   # uninterrupted
    pb, ob = make()
    run(pb, ob, grads[:2])
    sd = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in ob.state_dict().items()}     # as torch.save / load would
    sd["state"] = {i: {k: (v.cpu().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in ob.state_dict()["state"].items()}
    pc = [torch.nn.Parameter(p.detach().clone()) for p in pb]
    oc = AdamW([dict(params=pc[:2], weight_decay=0.1), dict(params=pc[2:], weight_decay=0.0)], lr=1e-2)
    oc.load_state_dict(sd)
    run(pc, oc, grads[2:])
    for a, c in zip(pa, pc):
        assert torch.equal(a.detach(), c.detach())
    assert all(oc.state[p]["step"] == 4 for p in pc)
    run(pb, ob, grads[2:])                               # and load_state_dict into a LIVE optimizer that already has a plan
    import copy
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))   # deep copy: load_state_dict keeps same-device tensors by reference
    for p, x in zip(pb, grads[0]):
        p.grad = x.clone().cuda()
    for p, x in zip(pa, grads[0]):
        p.grad = x.clone().cuda()
    ob.clip_and_step(1.0); oa.clip_and_step(1.0)
    for a, b in zip(pa, pb):
        assert torch.equal(a.detach(), b.detach())
