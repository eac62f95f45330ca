"""CPU, gloo, world_size 2: the data-parallel glue (xpretrain_amd/distributed.py) reproduces the
single-process result on the concatenated global batch (SURVEY.md §5 "DP parity target"):
loss identical; feature-path gradients = global gradient / W (Horovod averaging); logit_scale gradient exact;
the no-collective gather backward equals the all-reduce Horovod would run."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import clipvip_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from xpretrain_amd import distributed as D
    D.init_from_env("gloo")
    torch.manual_seed(0)
    B, d = 3, 16
    # a tiny "encoder": features = normalize(x @ W); same weights on every rank, different data per rank
    Wv = torch.nn.Parameter(torch.randn(8, d)); Wt = torch.nn.Parameter(torch.randn(8, d))
    ls = torch.nn.Parameter(torch.tensor(2.0))
    D.broadcast_parameters(torch.nn.ParameterList([Wv, Wt, ls]))
    g = torch.Generator().manual_seed(100)
    xv_all, xt_all = torch.randn(world * B, 8, generator=g), torch.randn(world * B, 8, generator=g)
    xv, xt = xv_all[rank * B:(rank + 1) * B], xt_all[rank * B:(rank + 1) * B]
    red = D.GradBucketReducer([Wv, Wt, ls], bucket_mb=0.0001, average=True)
    assert len(red.buckets) >= 2
    for step in range(2):                       # second step checks zero_grad / bucket reuse
        red.zero_grad()
        assert Wv.grad is None
        vis = torch.nn.functional.normalize(xv @ Wv, dim=-1); txt = torch.nn.functional.normalize(xt @ Wt, dim=-1)
        gv, gt = D.gather_features(vis, txt, verify_identical=True)
        loss = O.nce_learnable_temp_loss(gv, gt, ls)
        loss.backward()
        red.synchronize()
    # single-process oracle on the global batch
    Wv2, Wt2, ls2 = Wv.detach().clone().requires_grad_(), Wt.detach().clone().requires_grad_(), ls.detach().clone().requires_grad_()
    ref = O.nce_learnable_temp_loss(torch.nn.functional.normalize(xv_all @ Wv2, dim=-1),
                                    torch.nn.functional.normalize(xt_all @ Wt2, dim=-1), ls2)
    ref.backward()
    ok = (torch.allclose(loss, ref, atol=1e-6)
          and torch.allclose(Wv.grad, Wv2.grad / world, atol=1e-6) and torch.allclose(Wt.grad, Wt2.grad / world, atol=1e-6)
          and torch.allclose(ls.grad, ls2.grad, atol=1e-6))
    # gradient accumulation (run_pretrain.py:373-382): first micro-step under no_sync, second one reduces the SUM of both
    single = [p.grad.clone() for p in (Wv, Wt, ls)]
    red.zero_grad()

    def micro():
        vis = torch.nn.functional.normalize(xv @ Wv, dim=-1); txt = torch.nn.functional.normalize(xt @ Wt, dim=-1)
        gv, gt = D.gather_features(vis, txt)
        O.nce_learnable_temp_loss(gv, gt, ls).backward()
    with red.no_sync():
        micro()
    micro()
    red.synchronize()
    ok = ok and all(torch.allclose(p.grad, 2 * g, atol=1e-6) for p, g in zip((Wv, Wt, ls), single))
    # a second un-guarded backward before synchronize() must raise, not silently drop its contribution
    red.zero_grad()
    micro()
    try:
        micro()
        ok = False
    except RuntimeError as e:
        ok = ok and "no_sync" in str(e)
    red.synchronize()
    # four features in one collective (the pre-training step, run_pretrain.py:344-353)
    f4 = [torch.full((B, 4), float(rank * 10 + i)) for i in range(4)]
    g4 = D.gather_packed(*f4)
    ok = ok and len(g4) == 4 and all(g.shape == (world * B, 4) and float(g[r * B, 0]) == r * 10 + i
                                     for i, g in enumerate(g4) for r in range(world))
    q.put((rank, bool(ok), float(loss), float(ref)))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_world2_matches_global_batch_oracle():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_single_process_passthrough():
    from xpretrain_amd import distributed as D
    a, b = torch.randn(2, 4), torch.randn(2, 4)
    ga, gb = D.gather_features(a, b)
    assert ga is a and gb is b and D.world_size() == 1 and D.rank() == 0


# ---------------------------------------------------------------------------------------------- resume: optimizer state
def _resume_worker(rank, world, port, tmpdir, q):
    """run_pretrain.py:231-232,269-291: only rank 0 finds a restore.pt (the other rank's directory is empty, standing for a
    rank that read a different / no generation); after E2E_TrainingRestorer every rank must hold rank 0's step counter,
    parameters and Adam moments."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import types
    from xpretrain_amd import distributed as D
    from xpretrain_amd.utils.load_save import E2E_TrainingRestorer
    D.init_from_env("gloo")
    torch.manual_seed(rank)                                   # different initial weights per rank on purpose
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    opt = torch.optim.AdamW([{"params": model[0].parameters(), "lr": 1e-2}, {"params": model[1].parameters(), "lr": 3e-3}])
    out = os.path.join(tmpdir, f"rank{rank}")
    os.makedirs(out, exist_ok=True)
    opts = types.SimpleNamespace(output_dir=out, save_steps_ratio=0.5, num_train_steps=4)
    if rank == 0:                                             # a previous run of rank 0: 3 steps, checkpoint at step 2
        g = torch.Generator().manual_seed(5)
        for _ in range(2):
            opt.zero_grad()
            model(torch.randn(4, 6, generator=g)).square().sum().backward()
            opt.step()
        from xpretrain_amd.utils import load_save as LS
        torch.save({"global_step": 2, "model_state_dict": LS.to_cpu_half(model.state_dict()),
                    "optim_state_dict": LS.to_cpu_half(opt.state_dict())}, os.path.join(out, "restore.pt"))
        with torch.no_grad():                                 # the live objects drift away from the checkpoint again
            for p in model.parameters():
                p.add_(1.0)
    dist.barrier()
    r = E2E_TrainingRestorer(opts, model, opt)                # rank 0 restores from file, rank 1 finds nothing; then sync_ranks
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]
                     + [opt.state[p][k].reshape(-1) for p in model.parameters() for k in ("exp_avg", "exp_avg_sq")]
                     + [torch.tensor([float(opt.state[p]["step"]) for p in model.parameters()])])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    lrs = [g["lr"] for g in opt.param_groups]
    q.put((rank, bool(same), r.global_step, lrs, float(flat.abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_resume_broadcasts_optimizer_state(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res                       # identical parameters, Adam moments and step counts
    assert [r[2] for r in res] == [2, 2], res                # both resume at rank 0's global_step
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4] and res[0][4] > 0, res


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it re-executes itself under torch.distributed.run and both ranks join
    the process group (gloo here: no GPU); --launch-check stops before any GPU work."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world_size"] == 2 and d["n_ranks_seen"] == 2 and d["n_gpus"] == 2, d


def _layout_worker(rank, world, port, q):
    """GradBucketReducer(layout_groups=...): groups are contiguous inside one bucket in the given order, published as gradient
    sinks; a producer that writes its gradients straight into the sink (as the native layer backward does on the GPU) gets the
    same averaged gradients as the pack-copy path, with .grad being the bucket views from the start."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from xpretrain_amd import distributed as D
    import xpretrain_amd.functional as XF
    D.init_from_env("gloo")
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 7, 3, 11, 2, 6, 4)]
    groups = [[ps[3], ps[1], ps[4]], [ps[6], ps[0]]]          # orders unrelated to the registration order
    ok = True
    for wire in (None, torch.bfloat16):
        red = D.GradBucketReducer(ps, bucket_mb=64 / (1 << 20), average=True, layout_groups=groups, wire_dtype=wire)   # 16 elements
        for g in groups:
            key = XF.grad_sink_key(g)
            sink = XF.GRAD_SINKS[key]
            ok = ok and sink.numel() == sum(p.numel() for p in g)
            b = red._bucket_of[id(g[0])]
            ok = ok and all(red._bucket_of[id(p)] is b for p in g)
            # the producer: writes the group's flat gradient into the sink, hands autograd views of it
            off = 0
            for p in g:
                sink[off:off + p.numel()] = (rank + 1) * torch.arange(p.numel(), dtype=torch.float32) + p.numel()
                p.grad = sink[off:off + p.numel()].view_as(p)
                off += p.numel()
            red._on_group(g[0])                                # ONE hook per group (its first parameter), credited at the next hook
            views = {id(q): v for q, v in zip(b["params"], b["views"])}
            ok = ok and all(p.grad.data_ptr() == views[id(p)].data_ptr() for p in g)
        for p in (ps[2], ps[5]):                               # parameters outside any group: the usual adopted tensors
            p.grad = torch.full_like(p, float(rank + 1))
            red._on_grad(p)
        red.synchronize()
        mean = (1 + world) / 2
        for g in groups:
            for p in g:
                want = mean * torch.arange(p.numel(), dtype=torch.float32) + p.numel()
                ok = ok and torch.allclose(p.grad, want, rtol=1e-2 if wire else 1e-6)
        ok = ok and all(torch.allclose(p.grad, torch.full_like(p, mean), rtol=1e-2 if wire else 1e-6) for p in (ps[2], ps[5]))
        red.zero_grad()
        red.remove()
        ok = ok and not XF.GRAD_SINKS
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


class _ThreeParamLayer(torch.autograd.Function):
    """stand-in for functional.EncoderLayerFn: ONE autograd node that returns the gradients of all its parameters together"""

    @staticmethod
    def forward(ctx, x, a, b, c):
        ctx.save_for_backward(x, a, b, c)
        return x * a + b + c.sum()

    @staticmethod
    def backward(ctx, g):
        x, a, b, c = ctx.saved_tensors
        return g * a, g * x, g.clone(), g.sum() * torch.ones_like(c)


def _group_hook_worker(rank, world, port, q):
    """The reducer registers ONE hook per layout group and credits the group at the next hook: run through the REAL autograd engine
    (a chain of one-node layers, a layer applied twice, loose parameters before / between / after), buckets of about one layer,
    against the plain per-parameter averaging."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from xpretrain_amd import distributed as D
    D.init_from_env("gloo")
    torch.manual_seed(0)
    n, L = 6, 5
    layers = [[torch.nn.Parameter(torch.randn(n)), torch.nn.Parameter(torch.randn(n)), torch.nn.Parameter(torch.randn(3))] for _ in range(L)]
    head, mid, tail = (torch.nn.Parameter(torch.randn(n)) for _ in range(3))
    params = [head] + [p for lay in layers[:2] for p in lay] + [mid] + [p for lay in layers[2:] for p in lay] + [tail]
    groups = [[lay[1], lay[0], lay[2]] for lay in layers]                 # group order != argument order

    def loss_of(x):
        h = x * head
        for i, (a, b, c) in enumerate(layers):
            h = _ThreeParamLayer.apply(h, a, b, c)
            if i == 1:
                h = h + mid
                h = _ThreeParamLayer.apply(h, *layers[0])               # layer 0 applied a second time
        return (h * tail).sum()
    torch.manual_seed(100)
    xs = [torch.randn(n) * (r + 1) for r in range(world)]                # (every rank can build every rank's input)
    want = None
    for r in range(world):
        for p in params:
            p.grad = None
        loss_of(xs[r]).backward()
        g = [p.grad.clone() for p in params]
        want = g if want is None else [a + b for a, b in zip(want, g)]
    want = [w / world for w in want]
    for p in params:
        p.grad = None
    segs = [params[:8], params[8:]]                                        # a bucket never spans two segments
    red = D.GradBucketReducer(params, bucket_mb=4 * (2 * n + 3) / (1 << 20), average=True, layout_groups=groups, segments=segs)
    ok = len(red._hooks) == L + 3 and len(red.buckets) >= 3
    first = {id(p) for p in segs[0]}
    ok = ok and all(len({id(p) in first for p in b["params"]}) == 1 for b in red.buckets)
    for step in range(2):                                               # two steps: the pending state resets
        loss_of(xs[rank]).backward()
        launched = sum(b["work"] is not None for b in red.buckets)
        red.synchronize()
        ok = ok and launched >= len(red.buckets) - 2                    # all but the last bucket(s) left during backward
        ok = ok and all(torch.allclose(p.grad, w, rtol=1e-5, atol=1e-6) for p, w in zip(params, want))
        red.zero_grad()
    # gradient accumulation: first micro-step under no_sync
    with red.no_sync():
        loss_of(xs[rank]).backward()
    loss_of(xs[rank]).backward()
    red.synchronize()
    ok = ok and all(torch.allclose(p.grad, 2 * w, rtol=1e-5, atol=1e-6) for p, w in zip(params, want))
    red.zero_grad(); red.remove()
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_group_hooks_through_the_autograd_engine():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_group_hook_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_layout_groups_and_gradient_sinks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_layout_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


# ---------------------------------------------------------------------------------------------- optimizer state: source rank untouched
def _optstate_worker(rank, world, port, q):
    """broadcast_optimizer_state must never rewrite the SOURCE rank's tensors (a strided moment, or a scalar tensor living on
    another device kind than its parameter, used to be replaced by zeros on rank 0 and the zeros broadcast)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from xpretrain_amd import distributed as D
    D.init_from_env("gloo")
    p = torch.nn.Parameter(torch.zeros(4, 3))
    opt = torch.optim.SGD([p], lr=0.1 * (rank + 1))
    ok = True
    if rank == 0:
        strided = torch.arange(12.0).reshape(3, 4).t()          # [4, 3], not contiguous (a preserve_format moment)
        opt.state[p] = {"exp_avg": strided, "step": torch.tensor(7.0), "tag": "x"}
        ptr = strided.data_ptr()
    else:
        opt.state[p] = {"exp_avg": torch.full((4, 3), -1.0).t().contiguous().t(), "stale": torch.ones(2)}
    D.broadcast_optimizer_state(opt, src=0)
    st = opt.state[p]
    want = torch.arange(12.0).reshape(3, 4).t()
    ok = ok and torch.equal(st["exp_avg"], want) and float(st["step"]) == 7.0 and st["tag"] == "x" and "stale" not in st
    ok = ok and abs(opt.param_groups[0]["lr"] - 0.1) < 1e-12
    if rank == 0:
        ok = ok and st["exp_avg"].data_ptr() == ptr and not st["exp_avg"].is_contiguous()      # the very same tensor, untouched
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_optimizer_state_keeps_the_source_state():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_optstate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res


def test_gradient_sink_is_claimed_once_per_backward():
    """functional._claim_sink: a layer applied twice in ONE backward graph (VidCLIP's second image / caption pass) may write the
    sink once; the second application gets a private buffer and autograd adds the two -- g1 + g2, not 2 * g2."""
    import xpretrain_amd.functional as XF
    w = torch.nn.Parameter(torch.ones(3))
    sink = torch.zeros(3)
    key = XF.grad_sink_key((w,))

    class Layer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, scale):
            ctx.scale = scale
            return x * w

        @staticmethod
        def backward(ctx, g):
            flat = XF._claim_sink(key, sink, 3, sink.device, (w,))
            if flat is None:
                flat = torch.empty(3)
            flat.copy_(torch.full((3,), float(ctx.scale)))       # the "kernel" writes this call's gradient
            return g, flat.view(3), None

    x = torch.ones(3, requires_grad=True)
    for step in range(2):                                        # claims must not leak from one backward() into the next
        w.grad = None
        (Layer.apply(x, w, 1.0).sum() + Layer.apply(x, w, 10.0).sum()).backward()
        assert torch.equal(w.grad, torch.full((3,), 11.0)), w.grad
    assert XF._claim_sink(key, sink, 3, sink.device, (w,)) is None      # outside backward(): never
    XF.release_grad_sinks()
