"""Data-parallel glue for one 8xMI355X node: one process per GPU, ``torch.distributed`` backend ``nccl``
(= RCCL over xGMI on ROCm).  Replaces the reference's Horovod usage (run_pretrain.py:226-232,344-353,379;
utils/distributed.py) with three pieces designed for point-to-point xGMI rather than NVSwitch:

* ``gather_features``  -- ONE packed all-gather of the [2,B,d] (video,text) features per step instead of two
  (16 KB messages are latency-bound), differentiable.  Backward takes the LOCAL slice of the
  incoming gradient and issues no collective: every rank computes the same loss from bit-identical gathered
  inputs, so Horovod's averaged all-reduce in allgather-backward is a numerical no-op (SURVEY.md §5/§8e;
  ``verify_identical=True`` runs the real all-reduce and asserts that).
* ``GradBucketReducer`` -- a few large flat fp32 buckets (layer-reverse order, ~64 MB each: few, big messages that
  keep all 7 xGMI links busy); when a bucket's last gradient arrives its gradients are packed with one
  multi-tensor copy and the all-reduce is launched asynchronously, overlapping the rest of backward.
  ``average=True`` reproduces ``hvd.DistributedOptimizer`` (grads / world_size).
* ``broadcast_parameters`` / ``broadcast_optimizer_state`` -- rank-0 -> all at start-up and after a ``restore.pt`` resume
  (hvd.broadcast_parameters / hvd.broadcast_optimizer_state, run_pretrain.py:231-232).

Everything is backend-agnostic (the CPU tests run it on ``gloo`` with world_size 2).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> int:
    """Initialise the default process group from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns local rank."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend)
    return local_rank


def reserve_cus_for_collectives(n_channels: Optional[int] = None) -> int:
    """Tell the GEMM planning how many CUs RCCL's kernels take away beside the backward pass.  RCCL's gfx950 collective kernel
    (rcclGenericKernel in the librccl.so this image ships: 256 threads, 261-280 VGPRs + 17-32 AGPRs, 19.7 KB LDS -- llvm-readelf on
    its unbundled code object) runs ONE wave per SIMD, so a channel's workgroup owns a CU for as long as a bucket all-reduce lasts
    and cannot share it with a 256x256-GEMM workgroup.  (Since round 5 the split-K planning fills at most 176 CUs anyway --
    csrc/gemm.hip::XP_SPLITK_FILL -- so the budget only binds below that; it is still reported and still the place to tell the library.)
    ``n_channels`` defaults to $NCCL_MAX_NCHANNELS (bench.py sets it for its ranks), else 16.  Returns
    the budget set.  No-op in a 1-rank run."""
    if world_size() == 1:
        return 256
    if n_channels is None:
        n_channels = int(os.environ.get("NCCL_MAX_NCHANNELS", "16"))
    budget = max(64, 256 - max(0, int(n_channels)))
    if torch.cuda.is_available():
        from . import functional as XF
        XF.set_cu_budget(budget)
    return budget


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class _AllGatherRows(torch.autograd.Function):
    """concat over ranks along dim 0 (rank order), differentiable -- ``hvd.allgather`` semantics."""

    @staticmethod
    def forward(ctx, x, verify_identical):
        W = world_size()
        ctx.verify = verify_identical
        ctx.n = x.shape[0]
        x = x.contiguous()
        out = torch.empty((W * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x)
        return out

    @staticmethod
    def backward(ctx, g):
        r = rank()
        if ctx.verify:    # debug switch: the collective Horovod would run; must equal the local gradient
            avg = g.clone()
            dist.all_reduce(avg)
            avg /= world_size()
            if not torch.allclose(avg, g, rtol=1e-5, atol=1e-7):
                raise RuntimeError("gathered-feature gradients differ across ranks: the no-collective backward "
                                   "of gather_features is not valid for this loss")
        return g[r * ctx.n:(r + 1) * ctx.n], None


# Test hook: run the collectives even in a 1-rank group (exercises the RCCL code path on a single GPU).
FORCE_COLLECTIVES = False


def _single() -> bool:
    return world_size() == 1 and not (FORCE_COLLECTIVES and dist.is_initialized())


def allgather(x: torch.Tensor, verify_identical: bool = False) -> torch.Tensor:
    if _single():
        return x
    return _AllGatherRows.apply(x, verify_identical)


def gather_packed(*feats: torch.Tensor, verify_identical: bool = False):
    """k feature matrices [B,d] -> k gathered matrices [W*B,d] with ONE collective (rank-major rows, differentiable).
    k = 2: the fine-tuning step (run_pretrain.py:344-345); k = 4: the pre-training step's video / subtitle / frame /
    caption features (run_pretrain.py:344-353, four hvd.allgather calls in the reference)."""
    if _single():
        return feats
    packed = torch.stack(feats, dim=1)                            # [B,k,d]: rows stay rank-major after the gather
    g = _AllGatherRows.apply(packed, verify_identical)            # [W*B,k,d]
    return tuple(g[:, i] for i in range(len(feats)))


def gather_features(vis: torch.Tensor, txt: torch.Tensor, verify_identical: bool = False):
    """(vis[B,d], txt[B,d]) -> (vis[W*B,d], txt[W*B,d]) with ONE collective."""
    return gather_packed(vis, txt, verify_identical=verify_identical)


# XPRETRAIN_DEBUG=no_comm: the reducer packs and tracks its buckets but skips the all-reduce calls (cost attribution on one GPU)
_DEBUG_NO_COMM = "no_comm" in os.environ.get("XPRETRAIN_DEBUG", "").split(",")


class _NoWork:
    def wait(self):
        return True


class GradBucketReducer:
    """Bucketed, backward-overlapped gradient all-reduce (the role of hvd.DistributedOptimizer + synchronize()).

    Gradients start each step as ``None``: autograd then *adopts* the tensors our backward kernels return (no
    per-parameter ``grad += new`` kernels, no zero-fill of 600 MB of gradient memory).  When the last gradient of
    a bucket arrives (autograd hook), the bucket's gradients are packed into its flat fp32 buffer with ONE
    multi-tensor copy, ``.grad`` is re-pointed at the flat views, and the bucket's all-reduce is launched
    asynchronously -- overlapping the rest of backward.  With world_size 1 nothing is copied at all.

    Gradient accumulation (run_pretrain.py:373-382, ``skip_synchronize`` on all but the last micro-step): run the first
    micro-steps under ``with reducer.no_sync():`` -- no hook counts, autograd accumulates into ``.grad`` as usual -- and
    the last one outside it: the buckets then pack and reduce the accumulated gradients.  A second backward without
    ``no_sync`` before ``synchronize()`` is an error (the in-flight collective would miss its contribution) and raises."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, average: bool = True,
                 group=None, layout_groups=None, wire_dtype=None, segments=None):
        """``layout_groups``: lists of parameters that must sit contiguously, in the given order, inside one bucket -- the flat
        gradient layout a producer writes in one piece (``functional.layer_grad_groups(model)``: the 16 parameters of an encoder
        layer in the order of the native layer backward's flat buffer).  With it the bucket storage is allocated up front and
        published through ``functional.GRAD_SINKS``: the layer backward writes its gradients straight into the bucket, the
        pack copy (598 MB per step at ViT-B/16 + text tower) disappears, ``.grad`` are views of the bucket from the start.
        ``wire_dtype=torch.bfloat16``: the collective runs on a bf16 copy of the bucket (half the bytes over xGMI) and the
        result is widened back into the fp32 bucket; default None = fp32 on the wire, Horovod's arithmetic.
        ``segments``: lists of parameters whose gradients are produced on ONE stream each (``tower_segments(model)``: the video
        tower on the caller's stream, the text tower on its side stream); a bucket never spans two segments, so a bucket is packed
        and handed to the collective on the stream that produced it and no stream ever waits for another one inside backward."""
        self.group = group
        self.average = average
        self.wire_dtype = wire_dtype
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.buckets = []
        self._bucket_of = {}
        self._active = not _single()
        cap = int(bucket_mb * (1 << 20)) // 4
        # units of placement: a layout group (kept together, internal order fixed) or a single parameter
        group_of, units, seen = {}, [], set()
        for gparams in layout_groups or ():
            gparams = [p for p in gparams if p.requires_grad]
            if gparams and all(any(p is q for q in self.params) for p in gparams):
                for p in gparams:
                    group_of[id(p)] = gparams
        for p in reversed(self.params):        # gradients become ready roughly in reverse parameter order
            if id(p) in seen:
                continue
            unit = group_of.get(id(p), [p])
            for q in unit:
                seen.add(id(q))
            units.append(unit)
        seg_of = {}
        for i, seg in enumerate(segments or ()):
            for p in seg:
                seg_of.setdefault(id(p), i)
        cur, cur_n, cur_seg = [], 0, None
        for unit in units:
            n = sum(p.numel() for p in unit)
            useg = seg_of.get(id(unit[0]), -1)
            if cur and (cur_n + n > cap or useg != cur_seg):
                self._make_bucket(cur)
                cur, cur_n = [], 0
            cur.extend(unit)
            cur_n += n
            cur_seg = useg
        if cur:
            self._make_bucket(cur)
        self._sinks = []
        if self._active and layout_groups:
            from . import functional as XF
            for b in self.buckets:
                self._ensure_flat(b)
            for gparams in layout_groups:
                gparams = [p for p in gparams if p.requires_grad]
                if not gparams or id(gparams[0]) not in group_of:
                    continue
                b = self._bucket_of[id(gparams[0])]
                i0 = next(i for i, q in enumerate(b["params"]) if q is gparams[0])
                off = sum(q.numel() for q in b["params"][:i0])
                n = sum(q.numel() for q in gparams)
                key = XF.grad_sink_key(gparams)
                XF.GRAD_SINKS[key] = b["flat"][off:off + n]
                self._sinks.append(key)
        # One hook per layout group, not per parameter: the 16 gradients of an encoder layer leave ONE autograd node together and their
        # AccumulateGrad nodes (top priority in the engine's ready queue) all run before any other node does, so the hook of the
        # group's first parameter stands for the group -- credited when the NEXT hook of any kind fires (by then the other 15 have
        # been accumulated; the hook itself may fire in the middle of its own group) or in synchronize().  ~45 Python hook calls per
        # step instead of ~400: with per-parameter hooks and events the backward of a data-parallel step was HOST-bound
        # (profiles/r04w_forced_collectives_timeline.txt: +7.6 ms of enqueue time per step).
        self._group_rep, grouped = {}, set()
        if self._active:
            for gparams in layout_groups or ():
                gparams = [p for p in gparams if p.requires_grad]
                if not gparams or id(gparams[0]) not in group_of or id(gparams[0]) in self._group_rep:
                    continue
                self._group_rep[id(gparams[0])] = (self._bucket_of[id(gparams[0])], tuple(gparams))
                grouped.update(id(q) for q in gparams[1:])
        self._pending = None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_group if id(p) in self._group_rep else self._on_grad)
                       for p in self.params if id(p) not in grouped] if self._active else []
        self._sync = True
        backend = dist.get_backend(group) if self._active else ""
        self._avg_in_collective = average and backend == "nccl"      # RCCL divides inside the reduction; gloo has no AVG

    def no_sync(self):
        """context manager: backward passes inside it only accumulate into ``.grad`` (no packing, no collective)"""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = old
        return ctx()

    def _make_bucket(self, ps):
        # `marks`: (stream, event) per hook of this step -- the event is recorded on the stream autograd produced those gradients on
        # (the text tower runs on a side stream), when the hook fires; `pool`: the events, reused step after step
        b = dict(flat=None, params=list(ps), views=None, ready=0, work=None, n=sum(p.numel() for p in ps), marks=[], pool=[])
        for p in ps:
            self._bucket_of[id(p)] = b
        self.buckets.append(b)

    def _ensure_flat(self, b):
        if b["flat"] is None:
            flat = torch.empty(b["n"], dtype=torch.float32, device=b["params"][0].device)
            views, off = [], 0
            for p in b["params"]:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            b["flat"], b["views"] = flat, views

    def _launch(self, b):
        """pack the bucket (one multi-tensor copy; parameters without a gradient contribute zeros) and all-reduce -- on the stream
        that produced the bucket's gradients, whichever hook happens to complete the bucket (the group hooks credit a layer one hook
        late, possibly from another tower's hook): a bucket of one stream waits for nothing; a mixed bucket runs on the stream of its
        last gradient and waits for the events of the others (recorded when each was credited: _flush_pending / _on_grad)."""
        self._ensure_flat(b)
        marks = b["marks"]
        if marks:
            target = marks[-1][0]
            with torch.cuda.stream(target):
                foreign = False
                for st, ev in marks:
                    if st.cuda_stream != target.cuda_stream:
                        target.wait_event(ev)
                        foreign = True
                self._pack_and_reduce(b, foreign, target)
        else:
            self._pack_and_reduce(b, False, None)

    def _pack_and_reduce(self, b, foreign, cur):
        src, dst = [], []
        for p, v in zip(b["params"], b["views"]):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad if p.grad.dtype == torch.float32 else p.grad.float())
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
            if foreign:
                # the sources may have been produced (and allocated) on another stream: re-pointing .grad below drops
                # their last reference, and the caching allocator would hand the block back to the PRODUCER stream's pool
                # at once -- where ongoing backward work could overwrite it while this copy is still queued
                for p, v in zip(b["params"], b["views"]):
                    if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                        p.grad.record_stream(cur)      # the ORIGINAL gradient (a .float() temporary is already cur's)
        for p, v in zip(b["params"], b["views"]):
            p.grad = v
        if _DEBUG_NO_COMM:          # probe (tools): everything but the collective itself
            b["work"] = _NoWork()
            return
        op = dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM
        if self.wire_dtype is not None and self.wire_dtype != torch.float32:
            if b.get("wire") is None:
                b["wire"] = torch.empty(b["n"], dtype=self.wire_dtype, device=b["flat"].device)
            b["wire"].copy_(b["flat"])
            b["work"] = dist.all_reduce(b["wire"], op=op, group=self.group, async_op=True)
        else:
            b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)

    def _mark(self, b, st):
        """(stream, event recorded NOW on it) for a bucket's CUDA gradients: every kernel issued on `st` so far is covered"""
        if st is None:
            return None
        i = len(b["marks"])
        while len(b["pool"]) <= i:
            b["pool"].append(torch.cuda.Event())
        ev = b["pool"][i]
        ev.record(st)
        return st, ev

    @staticmethod
    def _stream_of(p):
        return torch.cuda.current_stream(p.device) if p.is_cuda else None

    def _credit(self, b, n, mark):
        if b["work"] is not None or b["ready"] + n > len(b["params"]):
            raise RuntimeError("GradBucketReducer: a gradient arrived for a bucket whose all-reduce is already in flight -- "
                               "run all but the last micro-step of a gradient-accumulation step under reducer.no_sync(), "
                               "and call synchronize() + zero_grad() between optimizer steps")
        if mark is not None:
            b["marks"].append(mark)
        b["ready"] += n
        if b["ready"] == len(b["params"]):
            self._launch(b)

    def _flush_pending(self):
        if self._pending is not None:
            (b, gparams, st), self._pending = self._pending, None
            if any(q.grad is None for q in gparams):
                raise RuntimeError("GradBucketReducer: a layout group's gradients did not arrive together (the group hook assumes one "
                                   "autograd node produces all of them, as functional.EncoderLayerFn does)")
            # the group's event is recorded HERE, on the stream its first hook ran on: by now the AccumulateGrad nodes of all its
            # parameters have run (they outrank every other ready task), so the event also covers the `grad += new` kernels of a
            # gradient-accumulation step and the clones of non-stealable gradients -- an event recorded in the first hook would not,
            # and a mixed-stream bucket packed on the other tower's stream could read gradients still being accumulated
            self._credit(b, len(gparams), self._mark(b, st))

    def _on_group(self, p):
        """hook of a layout group's first parameter: the group is credited at the next hook / in synchronize()"""
        if not self._sync:
            return
        self._flush_pending()
        b, gparams = self._group_rep[id(p)]
        self._pending = (b, gparams, self._stream_of(p))

    def _on_grad(self, p):
        if not self._sync:
            return
        self._flush_pending()
        b = self._bucket_of[id(p)]
        self._credit(b, 1, self._mark(b, self._stream_of(p)))     # the hook runs on the stream autograd produced this gradient on

    def synchronize(self):
        """Wait for every bucket (launching those whose parameters did not all receive a gradient) and average."""
        if not self._active:
            return
        W = world_size()
        if self._sync:
            self._flush_pending()
        self._pending = None
        for b in self.buckets:
            if b["work"] is None:
                self._launch(b)
        for b in self.buckets:
            b["work"].wait()
            b["work"] = None
            b["marks"].clear()
            if b.get("wire") is not None:
                b["flat"].copy_(b["wire"])
            if self.average and not self._avg_in_collective:
                b["flat"].div_(W)
            b["ready"] = 0

    def zero_grad(self):
        """Use instead of optimizer.zero_grad(): gradients go back to ``None`` so the next backward's tensors are
        adopted without an accumulate kernel."""
        for p in self.params:
            p.grad = None
        self._pending = None
        for b in self.buckets:
            b["ready"] = 0
            b["marks"].clear()
        if self._sinks:
            from . import functional as XF
            XF.release_grad_sinks()

    def remove(self):
        for h in self._hooks:
            h.remove()
        if self._sinks:
            from . import functional as XF
            for key in self._sinks:
                XF.GRAD_SINKS.pop(key, None)
            self._sinks = []
            XF.release_grad_sinks()


def tower_segments(model: torch.nn.Module):
    """For ``GradBucketReducer(segments=...)``: [video tower + its projection, text tower + its projection] of a ``VidCLIP`` /
    ``CLIPModel`` -- the two parameter sets whose gradients ``CLIPModel.forward`` produces on two different streams (the text tower
    runs on a side stream, modeling/CLIP_ViP.py).  Parameters outside both (logit_scale) form buckets of their own."""
    clip = getattr(model, "clipmodel", model)
    segs = []
    for tower, proj in (("vision_model", "visual_projection"), ("text_model", "text_projection")):
        ps = []
        for name in (tower, proj):
            m = getattr(clip, name, None)
            if m is not None:
                ps.extend(m.parameters())
        if ps:
            segs.append(ps)
    return segs


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    if _single():
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src)
    from .functional import WEIGHTS          # the compute-dtype weight copies were made from the pre-broadcast values
    WEIGHTS.invalidate()


def broadcast_optimizer_state(optimizer: torch.optim.Optimizer, src: int = 0):
    """``hvd.broadcast_optimizer_state(optimizer, root_rank=0)`` (run_pretrain.py:232): after this call every rank holds rank
    ``src``'s optimizer state -- per-parameter tensors (``exp_avg`` / ``exp_avg_sq``), per-parameter scalars (``step``) and the
    param-group hyper-parameters (``lr`` ...).  Needed after a resume: a rank whose restore differed (or that did not restore)
    would otherwise step with its own Adam moments and silently diverge from step 1.  Parameters are matched by their
    position in ``param_groups`` (as ``Optimizer.state_dict`` does); state entries missing on a receiving rank are created,
    entries the source does not have are dropped.  Tensors travel packed per dtype in a few large broadcasts."""
    if _single():
        return
    params = [p for g in optimizer.param_groups for p in g["params"]]
    me = rank()
    if me == src:
        meta = {"groups": [{k: v for k, v in g.items() if k != "params"} for g in optimizer.param_groups], "state": []}
        for p in params:
            st = optimizer.state.get(p, {})
            ent = {}
            for k, v in st.items():
                if not torch.is_tensor(v):
                    ent[k] = ("value", v)
                elif v.device != p.device:
                    # e.g. torch.optim.Adam(W)'s ``step``: a CPU scalar tensor beside GPU parameters -- it travels by value
                    # inside this object broadcast and is re-created on ITS device kind, never on the parameter's
                    ent[k] = ("host_tensor", v.detach().cpu())
                else:
                    ent[k] = ("tensor", tuple(v.shape), str(v.dtype).replace("torch.", ""))
            meta["state"].append(ent)
        box = [meta]
    else:
        box = [None]
    dist.broadcast_object_list(box, src)
    meta = box[0]
    if len(meta["groups"]) != len(optimizer.param_groups) or len(meta["state"]) != len(params):
        raise RuntimeError("broadcast_optimizer_state: the optimizers of the ranks have different parameter groups")
    for g, mg in zip(optimizer.param_groups, meta["groups"]):
        g.update(mg)
    by_dtype = {}
    for p, ms in zip(params, meta["state"]):
        st = optimizer.state[p] if ms else optimizer.state.get(p)
        if not ms:
            if st is not None:
                optimizer.state.pop(p, None)
            continue
        for k in [k for k in st if k not in ms]:
            del st[k]
        for k, desc in ms.items():
            if desc[0] == "value":
                st[k] = desc[1]
            elif desc[0] == "host_tensor":
                if me != src:
                    st[k] = desc[1].clone()
            else:
                _, shape, dtype = desc
                dtype = getattr(torch, dtype)
                t = st.get(k)
                # the SOURCE rank's tensors are never touched (they ARE the state being sent); a receiver keeps its own tensor
                # whenever shape / dtype / device fit -- strides may differ (preserve_format moments of a strided parameter):
                # the payload is copied in through a contiguous staging buffer below
                if me != src and (not torch.is_tensor(t) or tuple(t.shape) != shape or t.dtype != dtype or t.device != p.device):
                    t = st[k] = torch.zeros(shape, dtype=dtype, device=p.device)
                by_dtype.setdefault((dtype, p.device), []).append(t)
    cap = 64 << 20
    for (dtype, device), ts in by_dtype.items():
        i = 0
        while i < len(ts):
            chunk, nbytes = [], 0
            while i < len(ts) and (not chunk or nbytes + ts[i].numel() * ts[i].element_size() <= cap):
                chunk.append(ts[i]); nbytes += ts[i].numel() * ts[i].element_size(); i += 1
            if me == src:
                flat = torch.cat([t.detach().reshape(-1) for t in chunk])          # (a copy: the source state stays as it is)
            else:
                flat = torch.empty(sum(t.numel() for t in chunk), dtype=dtype, device=device)
            dist.broadcast(flat, src)
            if me != src:
                off = 0
                with torch.no_grad():
                    for t in chunk:
                        t.copy_(flat[off:off + t.numel()].view(t.shape)); off += t.numel()
    if hasattr(optimizer, "_plan"):
        optimizer._plan = None        # optimization.AdamW caches moment addresses


def ranks_seen() -> int:
    """number of ranks that take part in the default group, measured by a collective (all-reduce of ones), not read from the
    environment: what ``bench.py`` prints as ``n_ranks_seen``."""
    if _single():
        return 1
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    one = torch.ones((), dtype=torch.int32, device=dev)
    dist.all_reduce(one)
    return int(one.item())
