"""Autograd layer over the HIP kernels: one ``torch.autograd.Function`` per *fused block* of the path.

The unit of fusion is the transformer layer (``EncoderLayerFn``), not the op: the forward chains
LN -> fused-QKV GEMM (+bias, q-scale) -> fused attention -> out-proj GEMM (+bias +residual) -> LN ->
fc1 GEMM (+bias +quick_gelu) -> fc2 GEMM (+bias +residual); the backward chains the matching dX / dW
GEMMs with the activation-derivative and residual-gradient adds folded into GEMM / LayerNorm epilogues.
All arithmetic happens in ``libxpretrain_hip.so`` (``hip_ops``); torch only owns the buffers and the graph.

Activations are 2-D ``[tokens, features]`` tensors in the compute dtype (bf16 by default); parameters stay
fp32 masters and are cast once per weight version (``WeightCache``).  Parameter gradients are fp32.
Reference call sites are cited per class (paths relative to /root/reference/CLIP-ViP/src).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from typing import Optional, Tuple

import torch

from . import _lib as L
from . import hip_ops as H


# ------------------------------------------------------------------------------------------ weights
class _Entry:
    __slots__ = ("ver", "buf", "refs")

    def __init__(self, ver, buf, refs):
        self.ver, self.buf, self.refs = ver, buf, refs


class WeightCache:
    """Compute-dtype copies of fp32 master weights.  An entry is refreshed when (a) the parameter's version counter
    moved (``load_state_dict`` / in-place edits), or (b) ANY optimizer stepped since the copy was made -- fused / foreach
    optimizers update parameters without bumping ``Tensor._version`` (observed with ``AdamW(fused=True)``), so a global
    optimizer-step hook advances ``generation``.  ``invalidate()`` forces a refresh by hand.  An optimizer that rewrites
    the copies itself (``optimization.AdamW``: the shadows are written by the update kernel) asks for their addresses
    with ``shadow_bindings`` and keeps them valid through ``invalidate(keep=...)``."""

    def __init__(self):
        self._c = {}
        self.generation = 0
        self.structure_version = 0      # bumped whenever a cached buffer is (re)allocated
        self.casts = 0                  # bumped whenever a cast kernel is launched (on the caller's current stream)
        self._hooked = False

    def _ensure_hook(self):
        if not self._hooked:
            from torch.optim.optimizer import register_optimizer_step_post_hook

            def hook(opt, args, kwargs):
                after = getattr(opt, "_xp_after_step", None)
                if after is not None:
                    after(self)
                else:
                    self.invalidate()
            register_optimizer_step_post_hook(hook)
            self._hooked = True

    def _ver(self, ws):
        return tuple((w._version, w.data_ptr(), self.generation) for w in ws)

    def invalidate(self, keep=None):
        """Every cached copy goes stale, except the entries in ``keep`` (a list from ``shadow_bindings``) which the
        caller has just rewritten from the current master values."""
        self.generation += 1
        for key, ent in keep or ():
            if self._c.get(key) is ent:
                ws = [r() for r in ent.refs]
                if all(w is not None for w in ws):
                    ent.ver = self._ver(ws)
                # the buffer was rewritten through a raw pointer: tell autograd, so that a graph which saved it for backward
                # (forward -> optimizer.step() -> backward) fails loudly instead of computing dX with the new weights
                torch.autograd.graph.increment_version(ent.buf)

    @staticmethod
    def _alive(ent, ws) -> bool:
        """ids are only unique among LIVE objects: an entry is valid only while its weak references still point at
        the very tensors being asked about (a freed model's id can be reused by a new parameter)."""
        return ent is not None and all(r() is w for r, w in zip(ent.refs, ws))

    def get(self, w: torch.Tensor, dtype) -> torch.Tensor:
        if dtype == torch.float32:
            return w.detach()
        return self.fused((w,), dtype, _single=True)

    def fused(self, ws, dtype, _single=False) -> torch.Tensor:
        """Row-concatenation of several [n_i, k] weights (or 1-D biases) in `dtype`: the fused QKV operand."""
        self._ensure_hook()
        key = (tuple(id(w) for w in ws), dtype)
        ver = self._ver(ws)
        ent = self._c.get(key)
        if not self._alive(ent, ws) or ent.ver != ver:
            rows = sum(w.shape[0] for w in ws) if not _single else ws[0].shape[0]
            shape = tuple(ws[0].shape) if _single else (rows,) + tuple(ws[0].shape[1:])
            reuse = ent is not None and tuple(ent.buf.shape) == shape and ent.buf.device == ws[0].device
            if not reuse:
                self.structure_version += 1
            buf = ent.buf if reuse else torch.empty(shape, dtype=dtype, device=ws[0].device)
            if reuse:
                torch.autograd.graph.increment_version(buf)      # in-place rewrite below (raw pointer): see invalidate()
            self.casts += 1
            if _single:
                H.cast(ws[0].detach(), dtype, out=buf)
            else:
                r = 0
                for w in ws:
                    H.cast(w.detach(), dtype, out=buf[r:r + w.shape[0]])
                    r += w.shape[0]
            ent = _Entry(ver, buf, tuple(weakref.ref(w) for w in ws))
            self._c[key] = ent
        return ent.buf

    def shadow_bindings(self, params):
        """For an optimizer that rewrites the cached copies in its update kernel: ``({id(p): (device address, XP dtype
        code)}, entries)`` over the cache entries ALL of whose source tensors are in ``params`` (at most one copy per
        parameter; any other copy simply goes stale at the step)."""
        ids = {id(p) for p in params}
        shadows, entries = {}, []
        for key, ent in self._c.items():
            ws = [r() for r in ent.refs]
            if any(w is None or id(w) not in ids or id(w) in shadows for w in ws) or not ent.buf.is_contiguous():
                continue
            code = {torch.bfloat16: 0, torch.float32: 1}.get(ent.buf.dtype)
            if code is None:
                continue
            off = 0
            for w in ws:
                shadows[id(w)] = (ent.buf.data_ptr() + off * ent.buf.element_size(), code)
                off += w.numel()
            entries.append((key, ent))
        return shadows, entries

    def clear(self):
        self._c.clear()
        self.structure_version += 1


WEIGHTS = WeightCache()


_SPLITS = {}


def _split_for(n_out: int, n_in: int, rows: int, dtype: torch.dtype, a_remap, slack: bool = False) -> int:
    """split-K factor for the weight-gradient GEMMs (kernel-family aware, decided by the library; memoised).  ``slack``: a launch
    nothing waits for soon (the first three dW GEMMs of a layer's backward, as csrc/layer.hip issues them): fewer, longer slabs."""
    key = (n_out, n_in, rows, dtype, a_remap, slack, int(L.lib().xp_get_cu_budget()))      # (the CU budget is part of the plan: ADVICE r5)
    s = _SPLITS.get(key)
    if s is None:
        s = _SPLITS[key] = H.gemm_auto_split(n_out, n_in, rows, dtype, lda=n_out, ldb=n_in, a_remap=a_remap, slack=slack)
    return s


def _wgrad(dy: torch.Tensor, x: torch.Tensor, rows: int, n_out: int, n_in: int, a_remap=(0, 0, 0), slack: bool = False) -> torch.Tensor:
    """dW[n_out, n_in] = dY[rows, n_out]^T . X[rows, n_in] in fp32 (both operands read k-strided, split-K)."""
    split = _split_for(n_out, n_in, rows, dy.dtype, tuple(a_remap), slack)
    dw = torch.empty((n_out, n_in), dtype=torch.float32, device=dy.device)
    if split == 1:
        H.gemm(dy, x, n_out, n_in, rows, a_kstrided=True, b_kstrided=True, lda=n_out, ldb=n_in, out=dw,
               out_dtype=torch.float32, a_remap=a_remap)
    else:
        ws = H.workspace(split * n_out * n_in * 4, dy.device, "splitk")
        H.gemm(dy, x, n_out, n_in, rows, a_kstrided=True, b_kstrided=True, lda=n_out, ldb=n_in, out=ws,
               split_k=split, a_remap=a_remap)
        H.splitk_reduce(ws, dw, splits=split)
    return dw


# ------------------------------------------------------------------------------------------ encoder layer, native sequencing
# LAYER_CALLS (default): one C-ABI call per layer pass (xp_encoder_layer_fwd / _bwd, csrc/layer.hip) -- the same entry points in
# the same order with the same arguments as the op-by-op path below, issued from native code; XPRETRAIN_DEBUG=op_by_op keeps the
# op-by-op path (tools/determinism_hunt.py fingerprints every call of it; tests compare the two paths bit for bit).
LAYER_CALLS = "op_by_op" not in L.DEBUG
_DT_CODE = {torch.bfloat16: L.XP_BF16, torch.float32: L.XP_F32}
_NULLCTX = contextlib.nullcontext()
_ES = {torch.bfloat16: 2, torch.float32: 4}
_PLANS = {}


def set_cu_budget(cus: int) -> None:
    """CUs the split-K planning of the weight-gradient GEMMs may fill (xp_set_cu_budget; 256 = the whole MI355X).  A data-parallel
    run lowers it by the workgroups its collectives keep resident beside the backward pass (distributed.reserve_cus_for_collectives).
    The memoised split factors and layer plans (workspace sizes depend on the split) are dropped."""
    L.check(L.lib().xp_set_cu_budget(int(cus)), "xp_set_cu_budget")
    _SPLITS.clear()
    _PLANS.clear()


def _a256(n: int) -> int:
    return (n + 255) & ~255


class _LayerPlan:
    """Sizes / offsets of one (shape, dtype) of encoder layer: the saved-activation arena of the forward, the flat
    parameter-gradient buffer of the backward, the workspace sizes."""

    def __init__(self, rows, D, Dff, B, S, heads, size, dtype):
        d = L.XpLayerDims()
        d.rows, d.D, d.Dff, d.B, d.S, d.heads = rows, D, Dff, B, S, heads
        d.M, d.N, d.L = size if size is not None else (0, 1, S)
        d.attn_mode = L.ATTN_PROXY if size is not None else L.ATTN_CAUSAL
        d.dtype, d.q_scale, d.ln_eps = _DT_CODE[dtype], (D // heads) ** -0.5, 1e-5
        self.dims = d
        es = _ES[dtype]
        # arena: bf16/fp32 activations then fp32 statistics, 256-byte aligned pieces
        self.off, o = {}, 0
        for name, n in (("h1", rows * D * es), ("qkv", rows * 3 * D * es), ("attn_o", rows * D * es), ("x2", rows * D * es),
                        ("h2", rows * D * es), ("pre", rows * Dff * es), ("act", rows * Dff * es), ("mean1", rows * 4),
                        ("rstd1", rows * 4), ("mean2", rows * 4), ("rstd2", rows * 4), ("stats", B * heads * S * 2 * 4)):
            self.off[name] = o
            o += _a256(n)
        self.arena_bytes = o
        lib = L.lib()
        self.fwd_ws = int(lib.xp_encoder_layer_fwd_workspace_bytes(C.byref(d)))
        self.bwd_ws = int(lib.xp_encoder_layer_bwd_workspace_bytes(C.byref(d)))
        # flat fp32 parameter gradients, in the order of EncoderLayerFn.forward's parameter arguments
        self.gsizes = [D, D, 3 * D * D, 3 * D, D * D, D, D, D, Dff * D, Dff, D * Dff, D]
        self.gnames = ["dln1_w", "dln1_b", "dwqkv", "dbqkv", "dwo", "dbo", "dln2_w", "dln2_b", "dw1", "db1", "dw2", "db2"]
        self.gtotal = sum(self.gsizes)


def _layer_plan(rows, D, Dff, B, S, heads, size, dtype) -> _LayerPlan:
    key = (rows, D, Dff, B, S, heads, size, dtype)
    p = _PLANS.get(key)
    if p is None:
        p = _PLANS[key] = _LayerPlan(*key)
    return p


def _native_ok(x, vecs, mats, pad_mask) -> bool:
    """The native layer calls hand raw device pointers to C: everything the op-by-op path validates per call (hip_ops._chk)
    is validated here in one place -- 1-D parameters contiguous fp32 on x's device, weight copies contiguous in the compute
    dtype, an int64 contiguous padding mask.  False -> the caller takes the op-by-op path, whose checks raise a TypeError
    that names the offending argument (a model cast with .bfloat16(), a bool mask, a strided parameter ...)."""
    dev, dt = x.device, x.dtype
    if not x.is_cuda or not x.is_contiguous():
        return False
    for v in vecs:
        if v.dtype != torch.float32 or v.device != dev or not v.is_contiguous():
            return False
    for m in mats:
        if m.dtype != dt or m.device != dev or not m.is_contiguous():
            return False
    if pad_mask is not None and (pad_mask.dtype != torch.int64 or pad_mask.device != dev or not pad_mask.is_contiguous()):
        return False
    return True


# ---- weights the optimizer is still writing (optimization.AdamW(overlap=...)) ---------------------------------------------------
# AdamW.step can run the update of the encoder layers >= K of both towers on a stream of its own, so that the HBM-bound optimizer
# pass overlaps the MFMA-bound first layers of the NEXT forward instead of standing between two steps.  It leaves the event here;
# CLIPEncoder.forward makes every stream that is about to run layer K wait for it.  Anything else that reads those parameters outside a
# model forward (state_dict(), a checkpoint writer, an evaluation with other code) calls join_late_weights() first -- the model's
# state_dict pre-hook and utils.load_save do.
LATE_WEIGHTS = {"event": None, "first_layer": 0}


def wait_late_weights(layer_index, *streams):
    """Before encoder layer ``layer_index`` runs on ``streams`` (default: the current stream)."""
    ev = LATE_WEIGHTS["event"]
    if ev is None or layer_index != LATE_WEIGHTS["first_layer"]:
        return
    for st in (streams or (torch.cuda.current_stream(),)):
        st.wait_event(ev)


def join_late_weights():
    """The current stream waits for the optimizer's late update (no-op when none is pending)."""
    ev = LATE_WEIGHTS["event"]
    if ev is not None:
        torch.cuda.current_stream().wait_event(ev)


# XPRETRAIN_FWD_SPLIT=0: the whole batch as one chain (A/B switch for the two half-batch chains of the video tower's forward)
FWD_SPLIT = os.environ.get("XPRETRAIN_FWD_SPLIT", "1") != "0"
FWD_SPLIT_MIN_ROWS = 8192          # below this a half-batch launch no longer fills the chip beside its twin
FWD_SPLIT_HOLD_BYTES = 48 << 30    # forward-only passes: the most a split may hold until its join (every layer's buffers; 4.2 GB at cfg #2)
# FWD_SPLIT_STREAM (module attribute; tools set it): which stream the second chain runs on.  "side": the library's weight-gradient stream, idle
# during the forward (xp_side_stream) -- the step then touches main + text tower + side = three streams, as before the split.  "own":
# a torch stream of its own.  Same speed on one GPU (15.75 vs 15.75-15.79 ms per step, profiles/r04w_in_step_ab_second_chain_stream.txt);
# the fewer streams a step touches the better it survives the streams a collective library or a prefetcher adds: a fifth stream
# touched by the step cost 5 ms per step on this runtime, a fourth nothing (profiles/r04u_stream_count_probe.txt).
FWD_SPLIT_STREAM = "side"
_SPLIT_STREAMS = {}


def second_chain_stream_mode() -> str:
    return "own" if FWD_SPLIT_STREAM == "own" else "side"


def _second_chain_stream(device):
    st = _SPLIT_STREAMS.get(device.index)
    if st is None:
        if second_chain_stream_mode() == "side":
            with torch.cuda.device(device):
                ptr = L.lib().xp_side_stream()
            if ptr:
                st = torch.cuda.ExternalStream(ptr, device=device)
        if st is None:
            st = torch.cuda.Stream(device=device)
        _SPLIT_STREAMS[device.index] = st
    return st


class ForwardSplit:
    """The video tower's training forward as TWO half-batch chains: samples are independent in the forward, every kernel of a layer
    is row-parallel (GEMM rows, LayerNorm rows, attention per sample), so the second half of the batch runs the same layer call on a
    second stream into the second half of the SAME full-batch buffers.  The chains do not wait for each other between layers -- except
    at a layer whose compute-dtype weight copy was (re)cast inside the call (the first pass after a checkpoint load or an optimizer
    that does not maintain the shadows; xp_adamw_step writes them itself, so training steps never re-cast): the second chain then waits
    for that cast.  One chain's HBM-bound kernels (LayerNorm, attention, GEMM epilogues) run beside the other's MFMA main loops, and each chain's 111- /
    333- / 444-tile GEMMs fill the CUs the other leaves idle.  The backward sees ordinary full-batch buffers.  Results are bit-identical
    to the single chain (the same kernels compute every row).  Nothing of a layer may be freed before the chains are joined: a
    training pass keeps its activations for the backward anyway; a forward-only pass (torch.no_grad: retrieval, inference) has the
    split HOLD every layer's buffers until join() (round 6; ~0.46 GB per layer at cfg #2).  Measured: profiles/r04r_split_batch_probe_gemm256_forced.txt
    (probe), r04s_in_step_ab_forward_two_chains.txt (the step: -0.46 ms)."""

    def __init__(self, device):
        self.stream = _second_chain_stream(device)
        self.main = torch.cuda.current_stream(device)
        self.stream.wait_stream(self.main)          # fork: everything the tower's input depends on
        self.held = []

    def hold(self, *tensors):
        """buffers the second chain reads or writes: alive until the chains are joined (the caching allocator knows only the stream
        that allocated them, and without an autograd node a layer's input and arena die when the next layer returns)"""
        self.held.extend(t for t in tensors if t is not None)

    def join(self):
        self.main.wait_stream(self.stream)
        self.held.clear()       # (freed on the main stream, behind the join)


def _layer_fwd_native(x, ln1_w, ln1_b, Wqkv, bqkv, Wo, bo, ln2_w, ln2_b, W1, b1, W2, b2, plan, pad_mask, keep_pre=True, side=None,
                      split=None):
    dev = x.device
    arena = torch.empty(plan.arena_bytes, dtype=torch.uint8, device=dev)
    x3 = torch.empty_like(x)
    side_out = side_x2 = None
    if side is not None:
        side_out = torch.empty_like(side)
        if keep_pre:            # training pass: the x2 side rows are kept for the backward's second LayerNorm
            side_x2 = torch.empty_like(side)
    d = plan.dims
    es, D, Dff = _ES[x.dtype], d.D, d.Dff
    base, off = arena.data_ptr(), plan.off
    video = d.attn_mode == L.ATTN_PROXY
    if split is None:
        parts = [(plan, 0, 0, None)]
    else:                       # two half-batch chains: (plan of half the batch, first row, first sample, stream)
        hp = _layer_plan(d.rows // 2, D, Dff, d.B // 2, d.S, d.heads, (d.M, d.N, d.L), x.dtype)
        if (d.rows // 2) % 4:   # the second chain's statistics / stream pointers are offset by rows/2 elements: keep them 16-byte aligned
            raise RuntimeError(f"ForwardSplit: (B/2)*S = {d.rows // 2} rows per chain must be a multiple of 4")
        parts = [(hp, 0, 0, None), (hp, d.rows // 2, d.B // 2, split.stream)]
    for pl, r0, b0, stream in parts:
        with (torch.cuda.stream(stream) if stream is not None else _NULLCTX):
            ws = H.workspace(pl.fwd_ws, dev, "layer_fwd")          # (per stream)
            a = L.XpLayerFwd()
            a.dims = pl.dims
            a.x, a.Wqkv, a.Wo, a.W1, a.W2 = x.data_ptr() + r0 * D * es, Wqkv.data_ptr(), Wo.data_ptr(), W1.data_ptr(), W2.data_ptr()
            a.ln1_w, a.ln1_b, a.bqkv, a.bo = ln1_w.data_ptr(), ln1_b.data_ptr(), bqkv.data_ptr(), bo.data_ptr()
            a.ln2_w, a.ln2_b, a.b1, a.b2 = ln2_w.data_ptr(), ln2_b.data_ptr(), b1.data_ptr(), b2.data_ptr()
            a.pad_mask = 0 if pad_mask is None else pad_mask.data_ptr()
            rD, rF = r0 * D * es, r0 * Dff * es
            a.h1, a.qkv, a.attn_o = base + off["h1"] + rD, base + off["qkv"] + 3 * rD, base + off["attn_o"] + rD
            a.x2, a.h2 = base + off["x2"] + rD, base + off["h2"] + rD
            a.pre, a.act, a.x3 = (base + off["pre"] + rF) if keep_pre else 0, base + off["act"] + rF, x3.data_ptr() + rD
            a.mean1, a.rstd1 = base + off["mean1"] + r0 * 4, base + off["rstd1"] + r0 * 4
            a.mean2, a.rstd2 = base + off["mean2"] + r0 * 4, base + off["rstd2"] + r0 * 4
            a.stats = base + off["stats"] + b0 * d.heads * d.S * 2 * 4
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
            if side is not None:
                so = (b0 * d.M if video else r0) * D * 4          # side rows: [B*M, D] (video) / [rows, D] (text), fp32
                a.side_in, a.side_out = side.data_ptr() + so, side_out.data_ptr() + so
                a.side_S, a.side_M = (d.S, d.M) if video else (1, 1)
                if side_x2 is not None:
                    a.side_x2 = side_x2.data_ptr() + so
            L.check(L.lib().xp_encoder_layer_fwd(C.byref(a), H._stream()), "xp_encoder_layer_fwd")
    return x3, arena, side_out, side_x2


# Gradient sinks (distributed.GradBucketReducer(layout_groups=...)): flat fp32 buffers -- slices of the reducer's all-reduce
# buckets -- keyed by the storage of a layer's 16 parameters.  The native layer backward writes its parameter gradients straight
# into the sink instead of a private buffer, so the bucket never needs a pack copy.  Used only while every parameter of the layer
# has ``.grad is None`` (gradient accumulation: the second micro-step must not overwrite what autograd is about to add to) and only
# by the first application of the layer in a backward pass (_claim_sink).
GRAD_SINKS = {}
# One backward pass may hand a sink to ONE layer call only.  A layer applied twice in one graph (VidCLIP.forward's second image /
# caption pass, VidCLIP.py:70-79) backpropagates through both applications before the AccumulateGrad nodes of the shared parameters
# run, so ``p.grad is None`` still holds at the second call: writing the sink again would overwrite the views the first call
# returned and autograd would sum two aliases (2 * g2 instead of g1 + g2).  The claim is keyed by autograd's graph-task id (one id
# per ``backward()`` call), so it needs no reset between steps; the second application gets a private buffer and autograd adds it.
_SINK_CLAIMS = {}


def grad_sink_key(params16):
    return tuple(p.data_ptr() for p in params16)


def _claim_sink(key, sink, numel, device, params):
    """The sink of ``key`` if this layer call may write its gradients straight into it, else None (private buffer)."""
    if sink is None or sink.numel() != numel or sink.device != device or not all(p.grad is None for p in params):
        return None
    task = torch._C._current_graph_task_id()
    if task < 0 or _SINK_CLAIMS.get(key) == task:      # outside a backward pass / second application in this pass
        return None
    _SINK_CLAIMS[key] = task
    return sink


def release_grad_sinks():
    """forget the per-backward sink claims (GradBucketReducer.zero_grad / remove; graph-task ids restart in a new process only,
    so this is hygiene, not a correctness requirement)"""
    _SINK_CLAIMS.clear()


def layer_grad_groups(model):
    """For ``GradBucketReducer(layout_groups=...)``: the 16 parameters of every CLIPEncoderLayer of ``model`` in the order of the
    native layer backward's flat gradient buffer (_LayerPlan.gnames; q / k / v rows of the fused dWqkv and dbqkv in that order)."""
    groups = []
    for m in model.modules():
        if all(hasattr(m, a) for a in ("self_attn", "mlp", "layer_norm1", "layer_norm2")):
            a, f = m.self_attn, m.mlp
            groups.append([m.layer_norm1.weight, m.layer_norm1.bias, a.q_proj.weight, a.k_proj.weight, a.v_proj.weight,
                           a.q_proj.bias, a.k_proj.bias, a.v_proj.bias, a.out_proj.weight, a.out_proj.bias,
                           m.layer_norm2.weight, m.layer_norm2.bias, f.fc1.weight, f.fc1.bias, f.fc2.weight, f.fc2.bias])
    return groups


def _layer_bwd_native(ctx, dx3, x, arena, ln1_w, ln2_w, Wqkv, Wo, W1, W2, pad_mask, plan, side=None, side_x2=None):
    dev = x.device
    need = ctx.needs_input_grad
    dx = torch.empty_like(x)
    flat = None
    if GRAD_SINKS and ctx.sink_key is not None and all(need[1:17]):
        flat = _claim_sink(ctx.sink_key, GRAD_SINKS.get(ctx.sink_key), plan.gtotal, dev, ctx.sink_params)
    if flat is None:
        flat = torch.empty(plan.gtotal, dtype=torch.float32, device=dev)
    parts = flat.split_with_sizes(plan.gsizes)
    ws = H.workspace(plan.bwd_ws, dev, "layer_bwd")
    a = L.XpLayerBwd()
    a.dims = plan.dims
    base, off = arena.data_ptr(), plan.off
    a.x, a.h1, a.qkv, a.attn_o, a.x2 = x.data_ptr(), base + off["h1"], base + off["qkv"], base + off["attn_o"], base + off["x2"]
    a.h2, a.pre, a.act = base + off["h2"], base + off["pre"], base + off["act"]
    a.Wqkv, a.Wo, a.W1, a.W2 = Wqkv.data_ptr(), Wo.data_ptr(), W1.data_ptr(), W2.data_ptr()
    a.ln1_w, a.ln2_w = ln1_w.data_ptr(), ln2_w.data_ptr()
    a.mean1, a.rstd1, a.mean2, a.rstd2 = base + off["mean1"], base + off["rstd1"], base + off["mean2"], base + off["rstd2"]
    a.stats = base + off["stats"]
    a.pad_mask = 0 if pad_mask is None else pad_mask.data_ptr()
    a.dx3, a.dx = dx3.data_ptr(), dx.data_ptr()
    # forward argument positions: 1,2 ln1 | 3..8 wq,bq,wk,bk,wv,bv | 9,10 wo,bo | 11,12 ln2 | 13,14 w1,b1 | 15,16 w2,b2
    want = dict(dln1_w=need[1], dln1_b=need[2], dwqkv=need[3] or need[5] or need[7], dbqkv=need[4] or need[6] or need[8],
                dwo=need[9], dbo=need[10], dln2_w=need[11], dln2_b=need[12], dw1=need[13], db1=need[14], dw2=need[15], db2=need[16])
    g = {}
    for name, t in zip(plan.gnames, parts):
        if want[name]:
            g[name] = t
            setattr(a, name, t.data_ptr())
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    if side is not None and side_x2 is not None:          # the forward's fp32 side rows of x and x2: read by the LayerNorm backward passes
        a.side_in, a.side_x2 = side.data_ptr(), side_x2.data_ptr()
        a.side_S, a.side_M = (plan.dims.S, plan.dims.M) if plan.dims.attn_mode == L.ATTN_PROXY else (1, 1)
    L.check(L.lib().xp_encoder_layer_bwd(C.byref(a), H._stream()), "xp_encoder_layer_bwd")
    D, Dff = plan.dims.D, plan.dims.Dff
    gw = lambda n, shape: g[n].view(shape) if n in g else None
    dwqkv, dbqkv = gw("dwqkv", (3 * D, D)), g.get("dbqkv")
    pick = lambda t, i, ok: t[i * D:(i + 1) * D] if (t is not None and ok) else None
    return (dx, g.get("dln1_w"), g.get("dln1_b"),
            pick(dwqkv, 0, need[3]), pick(dbqkv, 0, need[4]), pick(dwqkv, 1, need[5]), pick(dbqkv, 1, need[6]),
            pick(dwqkv, 2, need[7]), pick(dbqkv, 2, need[8]), gw("dwo", (D, D)), g.get("dbo"), g.get("dln2_w"), g.get("dln2_b"),
            gw("dw1", (Dff, D)), g.get("db1"), gw("dw2", (D, Dff)), g.get("db2"), None, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------ encoder layer
class EncoderLayerFn(torch.autograd.Function):
    """CLIPEncoderLayer.forward (modeling/CLIP_ViP.py:444-460) with CLIPAttention.forward2 (:332-381, video
    tower, ``size=(M,N,L)``) or CLIPAttention.forward (:266-330, text tower, causal + padding) and CLIPMLP
    (:392-396)."""

    @staticmethod
    def forward(ctx, x, ln1_w, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, ln2_w, ln2_b, w1, b1, w2, b2,
                B: int, S: int, heads: int, size: Optional[Tuple[int, int, int]], pad_mask: Optional[torch.Tensor],
                training: bool = True, side: Optional[torch.Tensor] = None, split: Optional["ForwardSplit"] = None):
        """``side`` (fp32, bf16 compute only): the fp32 side rows of ``x`` (SideRows, csrc/gemm_common.h) -- [B*M, D], the proxy rows of
        every sample, in the video tower; [B*S, D], the whole stream, in the text tower.  The call then returns ``(x3, side_out)``."""
        dt = x.dtype
        if side is not None and (dt != torch.bfloat16 or side.dtype != torch.float32 or not side.is_contiguous() or side.device != x.device
                                 or tuple(side.shape) != ((B * size[0] if size is not None else B * S), x.shape[1])):
            raise TypeError("EncoderLayerFn: side rows must be a contiguous fp32 [B*M, D] (video) / [B*S, D] (text) tensor beside a bf16 stream")
        rows, D = x.shape
        Dff = w1.shape[0]
        dh = D // heads
        if dh != 64:
            raise RuntimeError(f"xpretrain_amd attention kernels are built for head_dim 64, got {dh}")
        q_scale = dh ** -0.5
        casts0 = WEIGHTS.casts
        Wqkv = WEIGHTS.fused((wq, wk, wv), dt)
        bqkv = WEIGHTS.fused((bq, bk, bv), torch.float32)
        Wo, W1, W2 = WEIGHTS.get(wo, dt), WEIGHTS.get(w1, dt), WEIGHTS.get(w2, dt)
        if LAYER_CALLS and _native_ok(x, (ln1_w, ln1_b, bqkv, bo, ln2_w, ln2_b, b1, b2), (Wqkv, Wo, W1, W2), pad_mask):
            plan = _layer_plan(rows, D, Dff, B, S, heads, size, dt)
            if split is not None and (size is None or B % 2 or pad_mask is not None):
                split = None
            if split is not None and WEIGHTS.casts != casts0:       # a weight copy was (re)made on this stream just now: the second
                split.stream.wait_stream(torch.cuda.current_stream())   # chain must not read it before the cast has run
            x3, arena, side_out, side_x2 = _layer_fwd_native(x, ln1_w, ln1_b, Wqkv, bqkv, Wo, bo, ln2_w, ln2_b, W1, b1, W2, b2, plan,
                                                             pad_mask, keep_pre=training, side=side, split=split)
            if split is not None:
                split.hold(x, arena, x3, side, side_out, side_x2)
            ctx.save_for_backward(x, arena, ln1_w, ln2_w, Wqkv, Wo, W1, W2, pad_mask, side, side_x2)
            ctx.plan = plan
            if GRAD_SINKS:          # (data-parallel runs only) the layer's parameters in the flat gradient order
                ctx.sink_params = (ln1_w, ln1_b, wq, wk, wv, bq, bk, bv, wo, bo, ln2_w, ln2_b, w1, b1, w2, b2)
                ctx.sink_key = grad_sink_key(ctx.sink_params)
            else:
                ctx.sink_params, ctx.sink_key = (), None
            if side is not None:
                ctx.mark_non_differentiable(side_out)
                ctx.set_materialize_grads(False)       # no zero-filled "gradient" of the side rows in the backward
                return x3, side_out
            return x3
        ctx.plan = None
        sd = None if side is None else ((S, size[0]) if size is not None else (1, 1))
        side_x2 = None if side is None else torch.empty_like(side)
        side_out = None if side is None else torch.empty_like(side)

        lns = None if side is None else ((S, size[0], size[0]) if size is not None else (1, 1, 1))
        h1, mean1, rstd1 = H.layernorm_fwd(x, ln1_w, ln1_b, rows, D, x_side=side, side=lns)
        qkv = H.gemm(h1, Wqkv, rows, 3 * D, D, epilogue=L.EPI_BIAS_QSCALE, bias=bqkv, scale=q_scale, scale_cols=D)
        attn_o, stats = H.attn_fwd(qkv, B, S, heads, size=size, pad_mask=pad_mask)
        x2 = H.gemm(attn_o, Wo, rows, D, D, epilogue=L.EPI_BIAS_RESID, bias=bo.detach(), resid=x,
                    resid_side=side, out_side=side_x2, side=sd)
        h2, mean2, rstd2 = H.layernorm_fwd(x2, ln2_w, ln2_b, rows, D, x_side=side_x2, side=lns)
        pre = torch.empty((rows, Dff), dtype=dt, device=x.device) if training else None
        act = H.gemm(h2, W1, rows, Dff, D, epilogue=L.EPI_BIAS_GELU, bias=b1.detach(), aux=pre)
        x3 = H.gemm(act, W2, rows, D, Dff, epilogue=L.EPI_BIAS_RESID, bias=b2.detach(), resid=x2,
                    resid_side=side_x2, out_side=side_out, side=sd)

        if training:
            ctx.save_for_backward(x, ln1_w, mean1, rstd1, h1, qkv, attn_o, stats, x2, ln2_w, mean2, rstd2, h2, pre, act,
                                  Wqkv, Wo, W1, W2, pad_mask, side, side_x2)
        ctx.meta = (B, S, heads, size, q_scale, D, Dff)
        ctx.lns = lns
        if side is not None:
            ctx.mark_non_differentiable(side_out)
            ctx.set_materialize_grads(False)
            return x3, side_out
        return x3

    @staticmethod
    def backward(ctx, dx3, _dside=None):
        if dx3 is None:             # (set_materialize_grads(False): the layer output did not reach the loss)
            return (None,) * 25
        if ctx.plan is not None:
            x, arena, ln1_w, ln2_w, Wqkv, Wo, W1, W2, pad_mask, side, side_x2 = ctx.saved_tensors
            if dx3.dtype != x.dtype or dx3.device != x.device or dx3.shape != x.shape:
                raise TypeError(f"EncoderLayerFn.backward: incoming gradient is {dx3.dtype} {tuple(dx3.shape)} on {dx3.device}, "
                                f"expected {x.dtype} {tuple(x.shape)} on {x.device}")
            return _layer_bwd_native(ctx, dx3.contiguous(), x, arena, ln1_w, ln2_w, Wqkv, Wo, W1, W2, pad_mask, ctx.plan, side, side_x2)
        (x, ln1_w, mean1, rstd1, h1, qkv, attn_o, stats, x2, ln2_w, mean2, rstd2, h2, pre, act,
         Wqkv, Wo, W1, W2, pad_mask, side, side_x2) = ctx.saved_tensors
        B, S, heads, size, q_scale, D, Dff = ctx.meta
        rows = x.shape[0]
        dx3 = dx3.contiguous()
        defer = H.DeferredReduce(x.device)       # the 4 bias + 4 LayerNorm-parameter reductions finish in 2 launches
        # ---- MLP: x3 = x2 + fc2(quick_gelu(fc1(LN2(x2))))
        # fc1's bias gradient = column sums of dpre: taken from the epilogue of the GEMM that produces dpre
        # parameter gradients are skipped for frozen parameters (freeze_text_encoder, VidCLIP.py:96-103): positions in
        # forward's argument list -- 1,2 ln1 | 3..8 q,k,v | 9,10 out_proj | 11,12 ln2 | 13,14 fc1 | 15,16 fc2
        need = ctx.needs_input_grad
        if need[14]:
            dpre, db1 = H.gemm(dx3, W2, rows, Dff, D, b_kstrided=True, epilogue=L.EPI_GELU_BWD, resid=pre, colsum_defer=defer,
                                  colsum_name="db1")
        else:
            dpre, db1 = H.gemm(dx3, W2, rows, Dff, D, b_kstrided=True, epilogue=L.EPI_GELU_BWD, resid=pre), None
        dw2 = _wgrad(dx3, act, rows, D, Dff, slack=True) if need[15] else None
        dh2 = H.gemm(dpre, W1, rows, D, Dff, b_kstrided=True)
        dw1 = _wgrad(dpre, h2, rows, Dff, D, slack=True) if need[13] else None
        # out_proj's bias gradient = column sums of dx2, fc2's = column sums of dx3: both accumulated by the LayerNorm backward
        # that reads dx3 and writes dx2
        dx2, dln2_w, dln2_b, dbo, db2 = H.layernorm_bwd(dh2, x2, ln2_w, mean2, rstd2, rows, D, dres=dx3, defer=defer,
                                                        dx_colsum=True, dres_colsum=True, name="ln2", x_side=side_x2, side=ctx.lns)
        # ---- attention: x2 = x + out_proj(attn(qkv(LN1(x))))
        dattn = H.gemm(dx2, Wo, rows, D, D, b_kstrided=True)
        dwo = _wgrad(dx2, attn_o, rows, D, D, slack=True) if need[9] else None
        want_bqkv = need[4] or need[6] or need[8]
        if want_bqkv:       # the q/k/v bias gradients (column sums of dqkv) come out of the attention backward kernels
            dqkv, dbqkv = H.attn_bwd(qkv, attn_o, dattn, stats, B, S, heads, size=size, pad_mask=pad_mask, q_scale=q_scale,
                                     colsum_defer=defer, colsum_name="dbqkv")
        else:
            dqkv = H.attn_bwd(qkv, attn_o, dattn, stats, B, S, heads, size=size, pad_mask=pad_mask, q_scale=q_scale)
        dh1 = H.gemm(dqkv, Wqkv, rows, D, 3 * D, b_kstrided=True)
        if need[3] or need[5] or need[7]:
            dwqkv = _wgrad(dqkv, h1, rows, 3 * D, D)
            dwq, dwk, dwv = dwqkv[:D], dwqkv[D:2 * D], dwqkv[2 * D:]
        else:
            dwq = dwk = dwv = None
        if want_bqkv:
            dbq, dbk, dbv = dbqkv[:D], dbqkv[D:2 * D], dbqkv[2 * D:]
        else:
            dbq = dbk = dbv = None
        dx, dln1_w, dln1_b = H.layernorm_bwd(dh1, x, ln1_w, mean1, rstd1, rows, D, dres=dx2, defer=defer, name="ln1", x_side=side,
                                             side=ctx.lns)
        defer.flush()
        keep = lambda i, g: g if need[i] else None          # (LayerNorm parameter / out_proj bias sums ride on passes that run anyway)
        return (dx, keep(1, dln1_w), keep(2, dln1_b), dwq, dbq, dwk, dbk, dwv, dbv, dwo, keep(10, dbo), keep(11, dln2_w),
                keep(12, dln2_b), dw1, db1, dw2, keep(16, db2), None, None, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------ fp32 side rows (proxy tokens)
# XPRETRAIN_PROXY_FP32=1 (default): the video tower's M proxy tokens of every sample keep their residual stream in fp32 beside the
# bf16 rows ([B*M, D] fp32 "side rows": read by the LayerNorms, read / written by the residual GEMM epilogues).  Measured with the
# oracle's storage emulation (tools/residual_precision_experiment.py): |loss - fp32 reference| at configs[1] batch 2 drops from
# 4.4e-2 to 5.6e-3, the feature error halves -- the same as a fully fp32 residual stream, for 4 of 2356 rows.
PROXY_SIDE = os.environ.get("XPRETRAIN_PROXY_FP32", "1") != "0"


def proxy_side_rows(class_emb, added_cls, pos_w, B: int, M: int) -> torch.Tensor:
    """[B*M, D] fp32: class_embedding / added_cls + position_embedding[0] (CLIP_ViP.py:187-191), exact"""
    D = class_emb.shape[0]
    side = torch.empty((B * M, D), dtype=torch.float32, device=class_emb.device)
    return H.vip_proxy_rows(class_emb.detach(), added_cls.detach().contiguous(), pos_w.detach().contiguous(), side, B, M, M, D)


def with_side_rows(x: torch.Tensor, side: Optional[torch.Tensor], B: int, S: int) -> torch.Tensor:
    """hidden state for ``output_hidden_states``: with side rows, an fp32 copy of x whose proxy rows are the exact fp32 ones"""
    if side is None:
        return x
    D = x.shape[1]
    h = x.detach().float()
    h.view(B, S, D)[:, :side.shape[0] // B] = side.view(B, -1, D)      # (text tower: the side rows are the whole stream)
    return h


# False: always materialise the patch matrix instead of gathering it in the GEMM loader (module attribute: tests compare the two)
PATCH_GATHER = True


# ------------------------------------------------------------------------------------------ embeddings
class VisionEmbedFn(torch.autograd.Function):
    """CLIPVisionViPEmbeddings.forward (modeling/CLIP_ViP.py:168-197): conv patch embed as im2col + MFMA GEMM
    whose epilogue adds temporal[t] + position[1+l] and writes straight into token slot M + t*L + l; proxy rows
    (class_embedding / added_cls + position[0]) by a small fill kernel.  ``time_table`` is the [T,D] temporal
    table (already interpolated by the caller when T != temporal_size, :171-174)."""

    @staticmethod
    def forward(ctx, video, patch_w, class_emb, added_cls, pos_w, time_table, dtype, want_wgrad=True):
        """``want_wgrad``: the caller's ``torch.is_grad_enabled() and patch_w.requires_grad`` (grad mode is off inside
        Function.forward, and ``ctx.needs_input_grad`` ignores ``torch.no_grad()``): False lets the loader gather the patches."""
        Bv, T, Cc, Hh, Ww = video.shape
        D, _, P, _ = patch_w.shape
        gh, gw = Hh // P, Ww // P
        Lp = gh * gw
        M = 1 + added_cls.shape[0]
        S = M + T * Lp
        if pos_w.shape[0] != 1 + Lp:
            raise ValueError(f"position_embedding has {pos_w.shape[0]} rows but the frame has {Lp} patches (+1)")
        K = 3 * P * P
        frames = video.reshape(Bv * T, Cc, Hh, Ww).contiguous()
        if frames.dtype != torch.uint8:
            frames = frames.float()
        Wp = WEIGHTS.get(patch_w, dtype).view(D, K)
        x = torch.empty((Bv * S, D), dtype=dtype, device=video.device)
        pos = pos_w.detach().contiguous()
        tt = time_table.detach().contiguous().float()
        # The patch matrix is only materialised when the backward needs it (dW of the patch embedding reads it k-strided).  A pass
        # that does not (inference, a frozen patch embedding) lets the GEMM's operand loader gather the 8-pixel strips straight from
        # [BT,3,H,W] (XpGemmDesc::a_frames): same bf16 operand values, no im2col round trip.
        need_patches = bool(want_wgrad) and ctx.needs_input_grad[1]
        on_the_fly = (not need_patches and PATCH_GATHER and dtype == torch.bfloat16 and P % 8 == 0 and Ww % 8 == 0)
        patches = None
        if on_the_fly:
            H.gemm(None, Wp, Bv * T * Lp, D, K, out=x, epilogue=L.EPI_PATCH, tab1=tt, tab2=pos[1:], tab_L=Lp, c_remap=(T * Lp, S, M),
                   frames=frames, frame_patch=P, frame_norm=(H.CLIP_MEAN, H.CLIP_STD) if frames.dtype == torch.uint8 else None)
        else:
            if frames.dtype == torch.uint8:     # decoded frames: /255, CLIP mean/std and the cast happen inside the gather
                patches = H.im2col_u8(frames, P, dtype)
            else:
                patches = H.im2col(frames, P, dtype)
            H.gemm(patches, Wp, Bv * T * Lp, D, K, out=x, epilogue=L.EPI_PATCH, tab1=tt, tab2=pos[1:], tab_L=Lp,
                   c_remap=(T * Lp, S, M))
        H.vip_proxy_rows(class_emb.detach(), added_cls.detach().contiguous(), pos, x, Bv, S, M, D)
        ctx.save_for_backward(patches)
        ctx.meta = (Bv, T, Lp, M, S, D, K, tuple(patch_w.shape), added_cls.shape[0])
        return x

    @staticmethod
    def backward(ctx, dx):
        (patches,) = ctx.saved_tensors
        Bv, T, Lp, M, S, D, K, wshape, nadd = ctx.meta
        dx = dx.contiguous()
        d_class, d_added, d_pos, d_time = H.vip_embed_bwd(dx, Bv, M, T, Lp, D)
        dwp = None if patches is None else _wgrad(dx, patches, Bv * T * Lp, D, K, a_remap=(T * Lp, S, M)).view(wshape)
        return None, dwp, d_class, d_added[:nadd], d_pos, d_time, None, None


class TextEmbedFn(torch.autograd.Function):
    """CLIPTextEmbeddings.forward (modeling/CLIP_ViP.py:210-227): token_embedding[ids] + position_embedding[:Lt]."""

    @staticmethod
    def forward(ctx, ids, tok_w, pos_w, dtype):
        ids = ids.contiguous()
        x = H.text_embed_fwd(ids, tok_w.detach(), pos_w.detach(), dtype)
        ctx.save_for_backward(ids)
        ctx.meta = (tok_w.shape[0], pos_w.shape[0])
        return x

    @staticmethod
    def backward(ctx, dx):
        (ids,) = ctx.saved_tensors
        d_tok, d_pos = H.text_embed_bwd(ids, dx.contiguous(), *ctx.meta)
        return None, d_tok, d_pos, None


# ------------------------------------------------------------------------------------------ small blocks
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm (pre_layrnorm :881, post_layernorm :893, final_layer_norm :772)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, x_side=None, side=None, want_y_side=False):
        """``x_side`` (fp32) with ``side = (S, M, stride)``: rows r with r % S < M are read from x_side (fp32 side rows of the
        residual stream); ``want_y_side``: their fp32 result is returned too (``(y, y_side)``, y_side laid out like x_side)."""
        rows, D = x.shape
        y_side = torch.empty_like(x_side) if want_y_side else None
        y, mean, rstd = H.layernorm_fwd(x, gamma.detach(), beta.detach(), rows, D, x_side=x_side, y_side=y_side, side=side)
        ctx.save_for_backward(x, gamma, mean, rstd, x_side)
        ctx.side = side
        if want_y_side:
            ctx.mark_non_differentiable(y_side)
            ctx.set_materialize_grads(False)
            return y, y_side
        return y

    @staticmethod
    def backward(ctx, dy, _dys=None):
        if dy is None:
            return (None,) * 6
        x, gamma, mean, rstd, x_side = ctx.saved_tensors
        rows, D = x.shape
        dx, dg, db = H.layernorm_bwd(dy.contiguous(), x, gamma.detach(), mean, rstd, rows, D, x_side=x_side, side=ctx.side)
        return dx, dg, db, None, None, None


class GatherRowsFn(torch.autograd.Function):
    """pooled[b] = x[b, idx[b]]  (EOT-argmax pooling :776; idx=None -> token 0, the class proxy, :892)."""

    @staticmethod
    def forward(ctx, x, idx, B, S):
        D = x.shape[1]
        ctx.save_for_backward(idx)
        ctx.meta = (B, S, D)
        return H.gather_rows(x, idx, B, S, D)

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, S, D = ctx.meta
        return H.scatter_rows(dout.contiguous(), idx, B, S, D), None, None, None


class ProjectionFn(torch.autograd.Function):
    """bias-free nn.Linear: visual_projection / text_projection (:1142,:1145)."""

    @staticmethod
    def forward(ctx, x, w):
        rows, K = x.shape
        N = w.shape[0]
        Wc = WEIGHTS.get(w, x.dtype)
        ctx.save_for_backward(x, Wc)
        return H.gemm(x, Wc, rows, N, K)

    @staticmethod
    def backward(ctx, dy):
        x, Wc = ctx.saved_tensors
        rows, K = x.shape
        N = Wc.shape[0]
        dy = dy.contiguous()
        dx = H.gemm(dy, Wc, rows, K, N, b_kstrided=True)
        dw = _wgrad(dy, x, rows, N, K)
        return dx, dw


class L2NormFn(torch.autograd.Function):
    """x / ||x||_2 (:1148-1149); fp32 unit-norm features out."""

    @staticmethod
    def forward(ctx, x):
        rows, D = x.shape
        y, inv = H.l2norm_fwd(x, rows, D)
        ctx.save_for_backward(y, inv)
        ctx.dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        rows, D = y.shape
        return H.l2norm_bwd(dy.contiguous().float(), y, inv, rows, D, ctx.dtype)


def _sim(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a . b^T in fp32 through xp_sim_matrix (csrc/loss.hip) -- the package launches no vendor GEMM"""
    a, b = a.contiguous().float(), b.contiguous().float()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L.check(L.lib().xp_sim_matrix(H._p(a), H._p(b), H._p(out), a.shape[0], b.shape[0], a.shape[1], H._stream()), "xp_sim_matrix")
    return out


class SimLogitsFn(torch.autograd.Function):
    """``logits_per_text = text_embeds @ image_embeds.T * logit_scale.exp()`` (CLIP_ViP.py:1151-1153) of ``CLIPModel.forward``:
    a [B, B] fp32 product that VidCLIP never reads (modeling/CLIP_ViP.py resolves it on access).  Forward and both input gradients
    are products of the same small-matrix kernel."""

    @staticmethod
    def forward(ctx, text, image, log_scale):
        scale = log_scale.detach().float().exp()
        logits = _sim(text, image) * scale
        ctx.save_for_backward(text, image, logits, scale)
        return logits

    @staticmethod
    def backward(ctx, g):
        text, image, logits, scale = ctx.saved_tensors
        g = g.contiguous().float()
        gs = g * scale
        dt = _sim(gs, image.float().t()).to(text.dtype) if ctx.needs_input_grad[0] else None         # [n, m] . [m, d]
        di = _sim(gs.t(), text.float().t()).to(image.dtype) if ctx.needs_input_grad[1] else None     # [m, n] . [n, d]
        dls = (g * logits).sum().reshape(()) if ctx.needs_input_grad[2] else None                     # d/d log_scale of s * e^ls
        return dt, di, dls


class NCELossFn(torch.autograd.Function):
    """NCELearnableTempLoss.forward (optimization/loss.py:134-141).  The kernel produces the loss and its
    gradients in one pass; backward only applies the incoming scalar."""

    @staticmethod
    def forward(ctx, vis, txt, log_scale):
        loss, dv, dt, dls = H.nce_loss(vis.contiguous().float(), txt.contiguous().float(),
                                       log_scale.detach().float().reshape(()))
        ctx.save_for_backward(dv, dt, dls)
        return loss

    @staticmethod
    def backward(ctx, g):
        dv, dt, dls = ctx.saved_tensors
        return dv * g, dt * g, dls * g


class VscFcLossFn(torch.autograd.Function):
    """NCELearnableTempLoss_vsc_fc.forward (optimization/loss.py:296-324); loss and gradients in one kernel pass."""

    @staticmethod
    def forward(ctx, vis, txt, img, cap, log_scale):
        loss, *grads = H.vsc_fc_loss(vis.contiguous().float(), txt.contiguous().float(), img.contiguous().float(),
                                     cap.contiguous().float(), log_scale.detach().float().reshape(()))
        ctx.save_for_backward(*grads)
        return loss

    @staticmethod
    def backward(ctx, g):
        return tuple(t * g for t in ctx.saved_tensors)


def _layer_params(layer):
    """the 16 parameters of a CLIPEncoderLayer in EncoderLayerFn's argument order.  The sub-module handles are cached on the layer
    (nn.Module.__getattr__ chains cost ~10 us per call otherwise); the Parameter objects are read from the modules' own tables
    every time, so a replaced Parameter is always seen."""
    cached = layer.__dict__.get("_xp_mods")
    if cached is not None:
        # all 8 sub-modules are re-checked by identity through plain dict lookups (nn.Module keeps children in _modules):
        # module surgery on ANY of them (adapters, replaced LayerNorms ...) rebuilds the table
        mods, tabs = cached
        lm = layer._modules
        a, m = lm["self_attn"]._modules, lm["mlp"]._modules
        live = (lm["layer_norm1"], a["q_proj"], a["k_proj"], a["v_proj"], a["out_proj"], lm["layer_norm2"], m["fc1"], m["fc2"])
        for x, y, t in zip(live, mods, tabs):
            if x is not y or x._parameters is not t:
                cached = None
                break
    if cached is None:
        a, m = layer.self_attn, layer.mlp
        mods = (layer.layer_norm1, a.q_proj, a.k_proj, a.v_proj, a.out_proj, layer.layer_norm2, m.fc1, m.fc2)
        tabs = tuple(x._parameters for x in mods)
        layer.__dict__["_xp_mods"] = (mods, tabs)
    return tuple(d[k] for d in tabs for k in ("weight", "bias"))


def encoder_layer(x, layer, B, S, heads, size, pad_mask, side=None, split=None):
    """Apply ``EncoderLayerFn`` with the parameters of a ``CLIPEncoderLayer`` module.  With ``side`` (the proxy rows of x in fp32)
    returns ``(x3, side_out)``.  ``split``: a ``ForwardSplit`` (the tower runs as two half-batch chains)."""
    # `training` argument: forward-only passes (torch.no_grad: retrieval / inference) skip the MLP pre-activation
    return EncoderLayerFn.apply(x, *_layer_params(layer), B, S, heads, size, pad_mask, torch.is_grad_enabled(), side, split)
