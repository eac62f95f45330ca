"""AdamW with the reference's semantics, executed as ONE multi-tensor HIP launch (plus one for the gradient norm).

Mirrors ``src/optimization/adamw.py:11-103`` (``AdamW(params, lr, betas, eps=1e-6, weight_decay, correct_bias)``,
``.step(closure)``; state keys ``step`` / ``exp_avg`` / ``exp_avg_sq`` as ``E2E_TrainingRestorer`` checkpoints them,
``utils/load_save.py:306-314``).  On top of the reference surface:

* ``step(max_grad_norm=...)`` / ``clip_and_step(max_norm)`` fuse ``clip_grad_norm_(params, max_norm)``
  (``tasks/run_video_retrieval.py:390-392``) into the update: the kernel scales the gradients on the fly, the norm
  is left in ``last_grad_norm`` (device scalar, no host sync).  The clipped gradients are not written back.
* the compute-dtype copies of the weights (``functional.WEIGHTS``) are rewritten by the same kernel, so no cast
  kernels run in the next forward.
* ``overlap_next_forward(model, first_late_layer=K)``: the update of the encoder layers >= K of both towers runs on a stream of its
  own, behind the gradient norm, and the NEXT forward waits for it in front of layer K (``functional.LATE_WEIGHTS``): the HBM-bound
  optimizer pass (4.5 GB at ViT-B/16 + text tower, 0.78 ms) overlaps the MFMA-bound first layers instead of standing between two
  steps.  Same kernel, same arguments, same results; readers of those parameters outside a model forward call
  ``functional.join_late_weights()`` (``VidCLIP.state_dict`` and ``utils.load_save`` do).

There is no eager fallback: without the HIP library the step raises.
"""
import ctypes as C
import math

import torch
from torch.optim import Optimizer

from .. import _lib as L
from .. import hip_ops as H


class AdamW(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        if lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter: {} - should be in [0.0, 1.0[".format(betas[1]))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(eps))
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))
        self.last_grad_norm = None      # device fp32 scalar after a clipped step
        self._plan = None
        self._late = None               # (ids of the late parameters, first late layer) -- overlap_next_forward()
        self._late_stream = None

    def overlap_next_forward(self, model, first_late_layer=2):
        """Run the update of ``encoder.layers[first_late_layer:]`` of the model's towers on a side stream (see the module docstring).
        ``first_late_layer=None`` switches it off."""
        self._plan = None
        if first_late_layer is None:
            self._late = None
            return self
        ids = set()
        for m in model.modules():
            layers = getattr(m, "layers", None)
            if isinstance(layers, torch.nn.ModuleList) and type(m).__name__ == "CLIPEncoder":
                for layer in layers[first_late_layer:]:
                    ids.update(id(p) for p in layer.parameters())
        self._late = (frozenset(ids), int(first_late_layer)) if ids else None
        return self

    # ------------------------------------------------------------------------------------------ planning
    def _active(self):
        act = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                act.append((gi, p))
        return act

    def _build_plan(self, act, cache):
        """Static part of a step: device tables, chunk maps, shadow bindings.  Rebuilt when the set of stepped
        parameters, their storage, or the weight cache's set of buffers changes."""
        dev = act[0][1].device
        shadows, entries = cache.shadow_bindings([p for _, p in act])
        launches, tensors = [], []
        for gi, p in act:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise TypeError("xpretrain_amd AdamW: parameters must be contiguous fp32 tensors on one device")
            st = self.state[p]
            if st["exp_avg"].dtype != torch.float32 or not st["exp_avg"].is_contiguous() or st["exp_avg"].device != dev:
                raise TypeError("xpretrain_amd AdamW: optimizer state must be contiguous fp32 tensors on the parameter's device")
            tensors.append((gi, p, st))
        late_ids = self._late[0] if self._late else frozenset()
        sets = [[t for t in tensors if id(t[1]) not in late_ids], [t for t in tensors if id(t[1]) in late_ids]]
        parts = [(tl[i0:i0 + L.XP_OPT_MAX_TENSORS], k == 1) for k, tl in enumerate(sets) for i0 in range(0, len(tl), L.XP_OPT_MAX_TENSORS)]
        n_total_chunks = 0
        for part, late in parts:
            tab = (L.XpAdamTensor * len(part))()
            cmap = []
            for j, (gi, p, st) in enumerate(part):
                sh = shadows.get(id(p))
                tab[j].p, tab[j].m, tab[j].v = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                tab[j].shadow = sh[0] if sh else 0
                tab[j].shadow_dtype = sh[1] if sh else 0
                tab[j].numel = p.numel()
                for c in range((p.numel() + L.XP_OPT_CHUNK - 1) // L.XP_OPT_CHUNK):
                    cmap += [j, c]
            raw = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8)
            launches.append(dict(part=part, table=raw.to(dev), n=len(part),
                                 chunk_map=torch.tensor(cmap, dtype=torch.int32).to(dev),
                                 n_chunks=len(cmap) // 2, chunk_base=n_total_chunks, late=late,
                                 grads=(C.c_void_p * len(part))(), grp=(C.c_uint8 * len(part))()))
            n_total_chunks += len(cmap) // 2
        return dict(key=self._plan_key(act, cache), sig=[(gi, p, p.data_ptr(), self.state[p], self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr())
                         for gi, p in act], sv=cache.structure_version,
                    launches=launches, n_chunks=n_total_chunks, entries=entries,
                    partials=torch.empty(n_total_chunks, dtype=torch.float32, device=dev),
                    norm=torch.zeros((), dtype=torch.float32, device=dev))

    def _plan_key(self, act, cache):
        # parameter storage, moment storage (load_state_dict replaces the state tensors) and the weight cache's buffers
        def moments(p):
            st = self.state.get(p)
            return (st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) if st and "exp_avg" in st else (0, 0)
        return (tuple((gi, id(p), p.data_ptr()) + moments(p) for gi, p in act), cache.structure_version, self._late)

    # ------------------------------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None, max_grad_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        act = self._active()
        if not act:
            return loss
        from ..functional import WEIGHTS
        plan = self._plan
        # steady state: the same parameters and moment tensors with the same storage as when the plan was built, and no change in the
        # weight cache's buffers (the same facts _plan_key() hashes, checked without building the key)
        if not (plan is not None and plan["sv"] == WEIGHTS.structure_version and len(act) == len(plan["sig"])
                and all(a[1] is s[1] and a[0] == s[0] and a[1].data_ptr() == s[2] and self.state.get(a[1]) is s[3]
                        and s[3]["exp_avg"].data_ptr() == s[4]
                        and s[3]["exp_avg_sq"].data_ptr() == s[5] for a, s in zip(act, plan["sig"]))):
            for _, p in act:                 # state initialisation (adamw.py:62-68) before the plan key looks at it
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data)
                    st["exp_avg_sq"] = torch.zeros_like(p.data)
            if plan is None or plan["key"] != self._plan_key(act, WEIGHTS):
                self._plan = self._build_plan(act, WEIGHTS)
            plan = self._plan
        clip = max_grad_norm is not None and max_grad_norm > 0
        stream = H._stream()
        lib = L.lib()
        # effective hyper-parameter groups: (param group, step count) pairs
        eff, eff_index = [], {}
        for la in plan["launches"]:
            for j, (gi, p, st) in enumerate(la["part"]):
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise TypeError("xpretrain_amd AdamW: gradients must be contiguous fp32 tensors")
                la["grads"][j] = g.data_ptr()
                st["step"] += 1
                k = (gi, st["step"])
                idx = eff_index.get(k)
                if idx is None:
                    group = self.param_groups[gi]
                    b1, b2 = group["betas"]
                    step_size = group["lr"]
                    if group["correct_bias"]:
                        step_size = step_size * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
                    idx = eff_index[k] = len(eff)
                    eff.append(L.XpAdamGroup(group["lr"], b1, b2, group["eps"], group["weight_decay"], step_size))
                la["grp"][j] = idx
        if len(eff) > L.XP_OPT_MAX_GROUPS:
            raise RuntimeError(f"xpretrain_amd AdamW: {len(eff)} distinct (group, step) pairs in one step (max "
                               f"{L.XP_OPT_MAX_GROUPS}); step parameters with diverging step counts separately")
        groups = (L.XpAdamGroup * len(eff))(*eff)
        if clip:
            for la in plan["launches"]:
                part = plan["partials"][la["chunk_base"]:]
                L.check(lib.xp_grad_sqnorm_partials(la["table"].data_ptr(), la["chunk_map"].data_ptr(), la["n_chunks"],
                                                    la["grads"], la["n"], part.data_ptr(), stream),
                        "xp_grad_sqnorm_partials")
        from .. import functional as XF
        XF.join_late_weights()                      # (a previous step's late update nobody waited for: two steps without a forward)
        late = [la for la in plan["launches"] if la["late"]]
        overlap = bool(late) and not torch.cuda.is_current_stream_capturing()
        dev = plan["norm"].device

        def run(la, st):
            L.check(lib.xp_adamw_step(la["table"].data_ptr(), la["chunk_map"].data_ptr(), la["n_chunks"], la["grads"],
                                      la["grp"], la["n"], groups, len(eff),
                                      plan["partials"].data_ptr() if clip else None, plan["n_chunks"] if clip else 0,
                                      float(max_grad_norm) if clip else 0.0,
                                      plan["norm"].data_ptr() if clip else None, st), "xp_adamw_step")
        if overlap:
            # first in program order: beside the early launches and whatever the main stream enqueues next.  (Launching it BEHIND the
            # early launches -- so that those run alone at full bandwidth -- measured the same: 14.385 vs 14.369 ms per step,
            # profiles/r06s_in_step_ab_optimizer_overlap_order.txt)
            self._launch_late(plan, late, run, dev, XF)
        for la in plan["launches"]:
            if not (overlap and la["late"]):
                run(la, stream)
        self.last_grad_norm = plan["norm"] if clip else None
        self._wrote = plan["entries"]
        return loss

    def state_dict(self):
        from .. import functional as XF
        XF.join_late_weights()           # the moments of the late layers may still be in flight on the optimizer's stream
        return super().state_dict()

    def _launch_late(self, plan, late, run, dev, XF):
        """the update of the late parameters on the optimizer's own stream, behind everything the current stream has enqueued so far
        (the gradients, the norm partials and -- when called after them -- the early launches); leaves the event the next forward
        waits for in functional.LATE_WEIGHTS"""
        if self._late_stream is None or self._late_stream.device != dev:
            self._late_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        self._late_stream.wait_stream(main)
        with torch.cuda.stream(self._late_stream):
            seen = set()
            for la in late:
                for _, p, _ in la["part"]:          # the caching allocator must not hand a freed gradient to the next forward
                    sp = p.grad.untyped_storage().data_ptr()      # while this stream still reads it (zero_grad(set_to_none=True))
                    if sp not in seen:
                        seen.add(sp)
                        p.grad.record_stream(self._late_stream)
                run(la, self._late_stream.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(self._late_stream)
        XF.LATE_WEIGHTS["event"], XF.LATE_WEIGHTS["first_layer"] = ev, self._late[1]

    def load_state_dict(self, state_dict):
        from .. import functional as XF
        XF.join_late_weights()
        super().load_state_dict(state_dict)
        self._plan = None                # the moment tensors were replaced

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._plan = None

    def clip_and_step(self, max_norm):
        """``clip_grad_norm_(params, max_norm)`` + ``step()`` in one pass; returns the (device) gradient norm."""
        self.step(max_grad_norm=max_norm)
        return self.last_grad_norm

    def _xp_after_step(self, cache):
        """Called by the weight cache's optimizer hook instead of a blanket invalidation: every cached copy goes stale
        except the ones this step rewrote."""
        wrote = getattr(self, "_wrote", None)
        self._wrote = None
        cache.invalidate(keep=wrote)
