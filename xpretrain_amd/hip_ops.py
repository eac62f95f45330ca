"""Tensor-level wrappers over the C ABI (no autograd here -- see ``functional.py``).

Every function takes CUDA(HIP) tensors, allocates outputs with torch (device memory is
PyTorch's job), passes raw ``data_ptr()``s plus the current HIP stream to
``libxpretrain_hip.so`` and raises ``RuntimeError`` with ``xp_last_error()`` on failure.
Nothing here computes with torch ops.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L

_DT = {torch.bfloat16: L.XP_BF16, torch.float32: L.XP_F32}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}; xpretrain_amd computes in bfloat16 or float32")


def _chk(t: torch.Tensor, name: str, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor -- xpretrain_amd has no CPU path")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


_ws_cache = {}


def workspace(nbytes: int, device, tag: str = "ws") -> torch.Tensor:
    """Grow-only per-(device, stream, tag) scratch buffer.  Kernels on one stream execute in
    order, so consecutive users of the same tag never overlap."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream, tag)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# --------------------------------------------------------------------------------------- GEMM
def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, *, lda=None, ldb=None, out=None, ldc=None,
         a_kstrided=False, b_kstrided=False, out_dtype=None, epilogue=L.EPI_NONE, bias=None, scale=1.0,
         scale_cols=0, resid=None, ldr=None, aux=None, ldaux=None, tab1=None, tab2=None, tab_L=0,
         a_remap=(0, 0, 0), c_remap=(0, 0, 0), split_k=1, out_rows=None, colsum_defer=None, colsum_name="gemm_colsum",
         resid_side=None, out_side=None, side=None, frames=None, frame_patch=0, frame_norm=None):
    """C[M,N] = epilogue(sum_k A(m,k) B(n,k)); see include/xpretrain_hip.h::XpGemmDesc.

    ``frames`` ([BT,3,H,W] fp32 or uint8, contiguous) with ``frame_patch`` = P: A is the patch matrix of the frames, gathered by the
    operand loader (no im2col pass; pass ``A=None``); uint8 frames need ``frame_norm = (mean3, std3)``.

    ``colsum_defer`` (a DeferredReduce): also return the column sums of C (a bias gradient), as ``(C, colsum)``; the
    sums come out of the GEMM epilogue where the library supports it, otherwise from a separate pass over C; either
    way they are final only after ``colsum_defer.flush()``."""
    if frames is not None:
        _chk(frames, "frames"); _chk(B, "B", torch.bfloat16)
        if A is not None or frames.dim() != 4 or frames.shape[1] != 3 or not frames.is_contiguous() or \
                frames.dtype not in (torch.float32, torch.uint8) or (frames.dtype == torch.uint8 and frame_norm is None):
            raise TypeError("gemm: frames must be a contiguous [BT,3,H,W] fp32 / uint8 (with frame_norm) tensor and A must be None")
        A = B                                          # (dtype / device of the compute path; the descriptor's A stays NULL)
    else:
        _chk(A, "A"); _chk(B, "B", A.dtype)
    out_dtype = out_dtype or A.dtype
    if split_k > 1:
        out = torch.empty((split_k, M, N), dtype=torch.float32, device=A.device) if out is None else out
        out_dtype = torch.float32
    elif out is None:
        out = torch.empty((out_rows or M, N), dtype=out_dtype, device=A.device)
    d = L.XpGemmDesc()
    d.A, d.B, d.C = (0 if frames is not None else A.data_ptr()), B.data_ptr(), out.data_ptr()
    if frames is not None:
        d.a_frames, d.a_frames_u8 = frames.data_ptr(), int(frames.dtype == torch.uint8)
        d.fr_H, d.fr_W, d.fr_P = frames.shape[2], frames.shape[3], int(frame_patch)
        if frame_norm is not None:
            for c in range(3):
                d.fr_mean[c], d.fr_std[c] = float(frame_norm[0][c]), float(frame_norm[1][c])
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else (M if a_kstrided else K)
    d.ldb = ldb if ldb is not None else (N if b_kstrided else K)
    d.ldc = ldc if ldc is not None else N
    d.a_kstrided, d.b_kstrided = int(a_kstrided), int(b_kstrided)
    d.in_dtype, d.out_dtype = _dt(A), _DT[out_dtype]
    d.epilogue, d.split_k = epilogue, split_k
    d.a_grp, d.a_grp_stride, d.a_off = a_remap
    d.c_grp, d.c_grp_stride, d.c_off = c_remap
    d.bias = 0 if bias is None else _chk(bias, "bias", torch.float32).data_ptr()
    d.scale, d.scale_cols = float(scale), scale_cols
    d.resid = 0 if resid is None else _chk(resid, "resid", A.dtype).data_ptr()
    d.ldr = ldr if ldr is not None else N
    d.aux = 0 if aux is None else _chk(aux, "aux", out_dtype).data_ptr()       # BIAS_GELU without aux: forward-only
    d.ldaux = ldaux if ldaux is not None else N
    d.tab1 = 0 if tab1 is None else _chk(tab1, "tab1", torch.float32).data_ptr()
    d.tab2 = 0 if tab2 is None else _chk(tab2, "tab2", torch.float32).data_ptr()
    d.tab_L = tab_L
    if resid_side is not None or out_side is not None:       # fp32 side rows of the residual stream: side = (S, M)
        _chk(resid_side, "resid_side", torch.float32); _chk(out_side, "out_side", torch.float32)
        S_, M_ = side
        need = ((M - 1) // S_ * M_ + min(M_, S_)) * N
        if not (resid_side.is_contiguous() and out_side.is_contiguous()) or resid_side.numel() < need or out_side.numel() < need:
            raise ValueError("gemm: resid_side / out_side are too small or not contiguous")
        d.resid_side, d.out_side, d.side_S, d.side_M = resid_side.data_ptr(), out_side.data_ptr(), S_, M_
    if colsum_defer is None:
        L.check(L.lib().xp_gemm(C.byref(d), _stream()), "xp_gemm")
        return out
    nrows = L.lib().xp_gemm_colsum_rows(C.byref(d))
    if nrows == 0:
        L.check(L.lib().xp_gemm(C.byref(d), _stream()), "xp_gemm")
        return out, colsum_deferred(out, M, N, colsum_defer, ldx=d.ldc, name=colsum_name)
    cs = torch.empty(N, dtype=torch.float32, device=A.device)
    part = colsum_defer.slot(nrows * N * 4, colsum_name)
    d.colsum_partials = part.data_ptr()
    L.check(L.lib().xp_gemm(C.byref(d), _stream()), "xp_gemm")
    colsum_defer.add(part, 0, cs, nrows, N, N)
    return out, cs


def gemm_auto_split(M: int, N: int, K: int, dtype: torch.dtype, *, a_kstrided=True, b_kstrided=True, lda=None,
                    ldb=None, a_remap=(0, 0, 0), slack=False) -> int:
    """Split-K factor for a weight-gradient GEMM (see include/xpretrain_hip.h::xp_gemm_auto_split / _slack)."""
    d = L.XpGemmDesc()
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else (M if a_kstrided else K)
    d.ldb = ldb if ldb is not None else (N if b_kstrided else K)
    d.ldc = N
    d.a_kstrided, d.b_kstrided = int(a_kstrided), int(b_kstrided)
    d.in_dtype, d.out_dtype = _DT[dtype], _DT[torch.float32]
    d.a_grp, d.a_grp_stride, d.a_off = a_remap
    return int((L.lib().xp_gemm_auto_split_slack if slack else L.lib().xp_gemm_auto_split)(C.byref(d)))


def splitk_reduce(slabs: torch.Tensor, out: torch.Tensor, accumulate=False, splits=None) -> torch.Tensor:
    n = out.numel()
    if splits is None:
        if slabs.dtype != torch.float32:
            raise TypeError("splitk_reduce: pass `splits` explicitly when the slab buffer is an untyped workspace")
        splits = slabs.numel() // n
    L.check(L.lib().xp_splitk_reduce(_p(slabs), _p(out), n, splits, int(accumulate), _stream()),
            "xp_splitk_reduce")
    return out


def colsum(X: torch.Tensor, rows: int, cols: int, ldx=None, out=None, accumulate=False) -> torch.Tensor:
    _chk(X, "X")
    out = torch.empty(cols, dtype=torch.float32, device=X.device) if out is None else out
    nb = L.lib().xp_colsum_workspace_bytes(rows, cols)
    ws = workspace(nb, X.device, "colsum")
    L.check(L.lib().xp_colsum(_p(X), rows, cols, ldx or cols, _dt(X), _p(out), int(accumulate), _p(ws), ws.numel(),
                              _stream()), "xp_colsum")
    return out


class DeferredReduce:
    """Collects the second-level reductions of several bias / LayerNorm-parameter gradients and finishes them in two
    launches (xp_reduce_rows_batch).  Each deferred producer keeps its partial rows in its own workspace slot."""

    def __init__(self, device):
        self.device = device
        self.segs = []
        self._keep = []
        self._names = set()

    def add(self, part: torch.Tensor, part_offset: int, out: torch.Tensor, nrows: int, width: int, stride: int,
            accumulate=False):
        sg = L.XpReduceSeg()
        sg.in_ = part.data_ptr() + 4 * part_offset
        sg.out = out.data_ptr()
        sg.stride, sg.nrows, sg.width, sg.accumulate = stride, nrows, width, int(accumulate)
        self.segs.append(sg)
        self._keep.append((part, out))

    def slot(self, nbytes: int, name: str) -> torch.Tensor:
        """partial-row buffer of the producer called `name` (one slot per producer NAME: reordering the producers of a
        layer cannot alias two live partial buffers)"""
        if name in self._names:
            raise RuntimeError(f"DeferredReduce: two producers named {name!r} before flush()")
        self._names.add(name)
        return workspace(nbytes, self.device, "defer:" + name)

    def flush(self):
        if not self.segs:
            return
        lib = L.lib()
        for i in range(0, len(self.segs), L.XP_REDUCE_MAX_SEGS):
            chunk = self.segs[i:i + L.XP_REDUCE_MAX_SEGS]
            arr = (L.XpReduceSeg * len(chunk))(*chunk)
            nb = lib.xp_reduce_rows_batch_workspace_bytes(arr, len(chunk))
            ws = workspace(nb, self.device, "reduce_batch")
            L.check(lib.xp_reduce_rows_batch(arr, len(chunk), _p(ws), ws.numel(), _stream()), "xp_reduce_rows_batch")
        self.segs, self._keep, self._names = [], [], set()


def colsum_deferred(X: torch.Tensor, rows: int, cols: int, defer: DeferredReduce, ldx=None, name="colsum") -> torch.Tensor:
    """Bias gradient whose final reduction is finished by ``defer.flush()``."""
    _chk(X, "X")
    out = torch.empty(cols, dtype=torch.float32, device=X.device)
    chunks = L.lib().xp_colsum_partial_rows(rows, cols)
    part = defer.slot(chunks * cols * 4, name)
    L.check(L.lib().xp_colsum_partials(_p(X), rows, cols, ldx or cols, _dt(X), _p(part), part.numel(), _stream()),
            "xp_colsum_partials")
    defer.add(part, 0, out, chunks, cols, cols)
    return out


# --------------------------------------------------------------------------------------- LayerNorm
def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, rows: int, cols: int, ldx=None,
                  eps: float = 1e-5, x_side=None, y_side=None, side=None):
    """``side = (S, M, stride)`` with ``x_side`` / ``y_side`` (fp32 [.., cols]): rows r with r % S < M are read from x_side
    instead of x / also written in fp32 to y_side, at side row (r // S) * stride + r % S (xp_layernorm_fwd_side)."""
    _chk(x, "x"); _chk(gamma, "gamma", torch.float32); _chk(beta, "beta", torch.float32)
    y = torch.empty((rows, cols), dtype=x.dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    if x_side is None and y_side is None:
        L.check(L.lib().xp_layernorm_fwd(_p(x), ldx or cols, _p(gamma), _p(beta), _p(y), cols, _p(mean), _p(rstd),
                                         rows, cols, eps, _dt(x), _stream()), "xp_layernorm_fwd")
    else:
        S, M, stride = side
        for t, nm in ((x_side, "x_side"), (y_side, "y_side")):
            if t is not None:
                _chk(t, nm, torch.float32)
                if not t.is_contiguous() or t.numel() < ((rows - 1) // S * stride + min(M, S)) * cols:
                    raise ValueError(f"layernorm_fwd: {nm} is too small or not contiguous")
        L.check(L.lib().xp_layernorm_fwd_side(_p(x), ldx or cols, _p(gamma), _p(beta), _p(y), cols, _p(mean), _p(rstd),
                                              rows, cols, eps, _dt(x), _p(x_side), _p(y_side), S, M, stride, _stream()),
                "xp_layernorm_fwd_side")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, rows, cols, *, ldx=None, lddy=None, dres=None, dx=None, lddx=None,
                  dgamma=None, dbeta=None, accumulate=False, defer: "DeferredReduce" = None, dx_colsum=False, dres_colsum=False,
                  name="ln", x_side=None, side=None):
    """Returns (dx, dgamma, dbeta) -- plus colsum(dx) when ``dx_colsum``, plus colsum(dres) when ``dres_colsum`` (deferred mode only).
    ``x_side`` / ``side = (S, M, stride)``: the fp32 side rows of x the forward normalised (layernorm_fwd)."""
    _chk(dy, "dy"); _chk(x, "x", dy.dtype)
    S, M, stride = side if x_side is not None else (0, 0, 0)
    if x_side is not None:
        _chk(x_side, "x_side", torch.float32)
        if not x_side.is_contiguous() or x_side.numel() < ((rows - 1) // S * stride + min(M, S)) * cols:
            raise ValueError("layernorm_bwd: x_side is too small or not contiguous")
    dx = torch.empty((rows, cols), dtype=dy.dtype, device=dy.device) if dx is None else dx
    dgamma = torch.empty(cols, dtype=torch.float32, device=dy.device) if dgamma is None else dgamma
    dbeta = torch.empty(cols, dtype=torch.float32, device=dy.device) if dbeta is None else dbeta
    nb = L.lib().xp_layernorm_bwd_workspace_bytes(rows, cols)
    if defer is not None:     # parameter-gradient partial rows stay in their own slot until defer.flush()
        ws = defer.slot(nb, name)
        L.check(L.lib().xp_layernorm_bwd_partials_side(_p(dy), lddy or cols, _p(x), ldx or cols, _p(gamma), _p(mean), _p(rstd),
                                                       _p(dres), cols, _p(dx), lddx or cols, 2 if dres_colsum else int(dx_colsum),
                                                       rows, cols, _dt(dy), _p(x_side), S, M, stride, _p(ws), ws.numel(), _stream()),
                "xp_layernorm_bwd_partials_side")
        nrows = L.lib().xp_layernorm_bwd_partial_rows(rows)
        if dres_colsum and not (dx_colsum and dres is not None):
            raise ValueError("layernorm_bwd: dres_colsum needs dx_colsum and dres")
        pitch = (4 if dres_colsum else 3 if dx_colsum else 2) * cols
        defer.add(ws, 0, dgamma, nrows, cols, pitch, accumulate)
        defer.add(ws, cols, dbeta, nrows, cols, pitch, accumulate)
        if dx_colsum:
            dxs = torch.empty(cols, dtype=torch.float32, device=dy.device)
            defer.add(ws, 2 * cols, dxs, nrows, cols, pitch)
            if dres_colsum:
                drs = torch.empty(cols, dtype=torch.float32, device=dy.device)
                defer.add(ws, 3 * cols, drs, nrows, cols, pitch)
                return dx, dgamma, dbeta, dxs, drs
            return dx, dgamma, dbeta, dxs
        return dx, dgamma, dbeta
    if dx_colsum:
        raise ValueError("layernorm_bwd: dx_colsum needs a DeferredReduce")
    ws = workspace(nb, dy.device, "ln")
    L.check(L.lib().xp_layernorm_bwd_side(_p(dy), lddy or cols, _p(x), ldx or cols, _p(gamma), _p(mean), _p(rstd),
                                          _p(dres), cols, _p(dx), lddx or cols, _p(dgamma), _p(dbeta), int(accumulate),
                                          rows, cols, _dt(dy), _p(x_side), S, M, stride, _p(ws), ws.numel(), _stream()),
            "xp_layernorm_bwd_side")
    return dx, dgamma, dbeta


# --------------------------------------------------------------------------------------- probes
def probe_mfma_bf16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    c = torch.empty((64, 4), dtype=torch.float32, device=a.device)
    L.check(L.lib().xp_probe_mfma_bf16(_p(a), _p(b), _p(c), _stream()), "xp_probe_mfma_bf16")
    return c


def probe_mfma_f32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    c = torch.empty((64, 4), dtype=torch.float32, device=a.device)
    L.check(L.lib().xp_probe_mfma_f32(_p(a), _p(b), _p(c), _stream()), "xp_probe_mfma_f32")
    return c


def probe_tr16(data: torch.Tensor, lane_off: torch.Tensor) -> torch.Tensor:
    out = torch.empty((64, 4), dtype=torch.int16, device=data.device)
    L.check(L.lib().xp_probe_tr16(_p(data), _p(lane_off), _p(out), _stream()), "xp_probe_tr16")
    return out


# --------------------------------------------------------------------------------------- embeddings / glue
def im2col(video: torch.Tensor, P: int, dtype) -> torch.Tensor:
    """video fp32 [BT,3,H,W] -> [BT*(H/P)*(W/P), 3*P*P] in `dtype`."""
    _chk(video, "video", torch.float32)
    BT, Cc, H, W = video.shape
    assert Cc == 3 and video.is_contiguous()
    out = torch.empty((BT * (H // P) * (W // P), 3 * P * P), dtype=dtype, device=video.device)
    L.check(L.lib().xp_im2col(_p(video), _p(out), BT, H, W, P, _DT[dtype], _stream()), "xp_im2col")
    return out


# CLIP pixel statistics used by every CLIP-ViP transform (datasets/dataloader.py:212-213)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def im2col_u8(frames: torch.Tensor, P: int, dtype, mean=CLIP_MEAN, std=CLIP_STD) -> torch.Tensor:
    """decoded uint8 frames [BT,3,H,W] -> normalised patch matrix [BT*(H/P)*(W/P), 3*P*P] in `dtype`."""
    if frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous():
        raise TypeError("im2col_u8: frames must be a contiguous uint8 tensor on the GPU (no CPU path)")
    BT, Cc, H, W = frames.shape
    assert Cc == 3
    out = torch.empty((BT * (H // P) * (W // P), 3 * P * P), dtype=dtype, device=frames.device)
    m, sd = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    L.check(L.lib().xp_im2col_u8(_p(frames), m, sd, _p(out), BT, H, W, P, _DT[dtype], _stream()), "xp_im2col_u8")
    return out


def vip_proxy_rows(class_emb, added_cls, pos, x, B, S, M, D):
    L.check(L.lib().xp_vip_proxy_rows(_p(class_emb), _p(added_cls), _p(pos), _p(x), B, S, M, D, _dt(x), _stream()),
            "xp_vip_proxy_rows")
    return x


def vip_embed_bwd(dx, B, M, T, Lp, D, want_time=True):
    dev = dx.device
    d_class = torch.empty(D, dtype=torch.float32, device=dev)
    d_added = torch.empty((max(M - 1, 1), D), dtype=torch.float32, device=dev)
    d_pos = torch.empty((1 + Lp, D), dtype=torch.float32, device=dev)
    d_time = torch.empty((T, D), dtype=torch.float32, device=dev) if want_time else None
    nb = L.lib().xp_vip_embed_bwd_workspace_bytes(B, T, Lp, D)
    ws = workspace(nb, dev, "embed")
    L.check(L.lib().xp_vip_embed_bwd(_p(dx), _p(d_class), _p(d_added), _p(d_pos), _p(d_time), B, M, T, Lp, D, _dt(dx), 0,
                                     _p(ws), ws.numel(), _stream()), "xp_vip_embed_bwd")
    return d_class, d_added[: M - 1], d_pos, d_time


def text_embed_fwd(ids, tok, pos, dtype):
    _chk(ids, "ids", torch.int64)
    B, Lt = ids.shape
    vocab, D = tok.shape
    x = torch.empty((B * Lt, D), dtype=dtype, device=ids.device)
    L.check(L.lib().xp_text_embed_fwd(_p(ids), _p(tok), _p(pos), _p(x), B, Lt, D, vocab, _DT[dtype], _stream()),
            "xp_text_embed_fwd")
    return x


def text_embed_bwd(ids, dx, vocab, npos):
    B, Lt = ids.shape
    D = dx.shape[-1]
    d_tok = torch.empty((vocab, D), dtype=torch.float32, device=dx.device)
    d_pos = torch.zeros((npos, D), dtype=torch.float32, device=dx.device)
    L.check(L.lib().xp_text_embed_bwd(_p(ids), _p(dx), _p(d_tok), _p(d_pos), B, Lt, D, vocab, _dt(dx), 0, _stream()),
            "xp_text_embed_bwd")
    return d_tok, d_pos


def argmax_rows(ids):
    B, Lt = ids.shape
    idx = torch.empty(B, dtype=torch.int64, device=ids.device)
    L.check(L.lib().xp_argmax_rows(_p(ids), _p(idx), B, Lt, _stream()), "xp_argmax_rows")
    return idx


def gather_rows(x, idx, B, S, D):
    out = torch.empty((B, D), dtype=x.dtype, device=x.device)
    L.check(L.lib().xp_gather_rows(_p(x), _p(idx), _p(out), B, S, D, _dt(x), _stream()), "xp_gather_rows")
    return out


def scatter_rows(dout, idx, B, S, D):
    dx = torch.empty((B * S, D), dtype=dout.dtype, device=dout.device)
    L.check(L.lib().xp_scatter_rows(_p(dout), _p(idx), _p(dx), B, S, D, _dt(dout), _stream()), "xp_scatter_rows")
    return dx


def l2norm_fwd(x, rows, cols):
    y = torch.empty((rows, cols), dtype=torch.float32, device=x.device)
    inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    L.check(L.lib().xp_l2norm_fwd(_p(x), _p(y), _p(inv), rows, cols, _dt(x), _stream()), "xp_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, y, inv, rows, cols, dtype):
    _chk(dy, "dy", torch.float32)
    dx = torch.empty((rows, cols), dtype=dtype, device=dy.device)
    L.check(L.lib().xp_l2norm_bwd(_p(dy), _p(y), _p(inv), _p(dx), rows, cols, _DT[dtype], _stream()), "xp_l2norm_bwd")
    return dx


def cast(src: torch.Tensor, dtype, out=None) -> torch.Tensor:
    _chk(src, "src", torch.float32)
    out = torch.empty(src.shape, dtype=dtype, device=src.device) if out is None else out
    L.check(L.lib().xp_cast(_p(src), _p(out), src.numel(), _DT[dtype], _stream()), "xp_cast")
    return out


def cast_back(src: torch.Tensor, out=None, accumulate=False) -> torch.Tensor:
    out = torch.empty(src.shape, dtype=torch.float32, device=src.device) if out is None else out
    L.check(L.lib().xp_cast_back(_p(src), _p(out), src.numel(), _dt(src), int(accumulate), _stream()), "xp_cast_back")
    return out


# --------------------------------------------------------------------------------------- loss
def nce_loss(vis: torch.Tensor, txt: torch.Tensor, log_scale: torch.Tensor):
    """Returns (loss, d_vis, d_txt, d_log_scale) -- fp32 device tensors."""
    _chk(vis, "vis", torch.float32); _chk(txt, "txt", torch.float32); _chk(log_scale, "log_scale", torch.float32)
    n, d = vis.shape
    dev = vis.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dls = torch.empty((), dtype=torch.float32, device=dev)
    dv, dt = torch.empty_like(vis), torch.empty_like(txt)
    nb = L.lib().xp_nce_loss_workspace_bytes(n, d)
    ws = workspace(nb, dev, "loss")
    L.check(L.lib().xp_nce_loss(_p(vis), _p(txt), _p(log_scale), _p(loss), _p(dv), _p(dt), _p(dls), n, d, _p(ws),
                                ws.numel(), _stream()), "xp_nce_loss")
    return loss, dv, dt, dls


def vsc_fc_loss(vis, txt, img, cap, log_scale):
    """Returns (loss, d_vis, d_txt, d_img, d_cap, d_log_scale) -- fp32 device tensors."""
    for t, name in ((vis, "vis"), (txt, "txt"), (img, "img"), (cap, "cap"), (log_scale, "log_scale")):
        _chk(t, name, torch.float32)
    n, d = vis.shape
    if not (txt.shape == img.shape == cap.shape == vis.shape):
        raise ValueError(f"vsc_fc_loss: feature shapes differ: {tuple(vis.shape)}, {tuple(txt.shape)}, "
                         f"{tuple(img.shape)}, {tuple(cap.shape)}")          # loss.py:297 asserts text/cap only
    dev = vis.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dls = torch.empty((), dtype=torch.float32, device=dev)
    grads = [torch.empty_like(vis) for _ in range(4)]
    nb = L.lib().xp_vsc_fc_loss_workspace_bytes(n, d)
    ws = workspace(nb, dev, "loss")
    L.check(L.lib().xp_vsc_fc_loss(_p(vis), _p(txt), _p(img), _p(cap), _p(log_scale), _p(loss), _p(grads[0]),
                                   _p(grads[1]), _p(grads[2]), _p(grads[3]), _p(dls), n, d, _p(ws), ws.numel(),
                                   _stream()), "xp_vsc_fc_loss")
    return (loss, *grads, dls)


# --------------------------------------------------------------------------------------- attention
def _attn_ws(mode, B, H, M, N, Lp, device):
    nb = L.lib().xp_attn_workspace_bytes(mode, B, H, M, N, Lp)
    return workspace(nb, device, "attn")


def attn_fwd(qkv: torch.Tensor, B: int, S: int, H: int, *, size=None, pad_mask=None):
    """qkv [B*S, 3*H*64] (q pre-scaled).  size=(M,N,L) selects the video-proxy pattern, otherwise the
    causal(+padding) text pattern.  Returns (out [B*S, H*64], stats [B,H,S,2])."""
    _chk(qkv, "qkv")
    mode = L.ATTN_PROXY if size is not None else L.ATTN_CAUSAL
    M, N, Lp = size if size is not None else (0, 1, S)
    out = torch.empty((B * S, H * 64), dtype=qkv.dtype, device=qkv.device)
    stats = torch.empty((B, H, S, 2), dtype=torch.float32, device=qkv.device)
    if pad_mask is not None:
        _chk(pad_mask, "pad_mask", torch.int64)
    ws = _attn_ws(mode, B, H, M, N, Lp, qkv.device)
    L.check(L.lib().xp_attn_fwd(_p(qkv), 3 * H * 64, _p(out), H * 64, _p(stats), _p(pad_mask), mode, B, H, S, M, N, Lp,
                                _dt(qkv), _p(ws), ws.numel(), _stream()), "xp_attn_fwd")
    return out, stats


def attn_bwd(qkv, out, dout, stats, B, S, H, *, size=None, pad_mask=None, q_scale=1.0, colsum_defer=None,
             colsum_name="dbqkv"):
    """dqkv; with ``colsum_defer`` (a DeferredReduce) also the column sums of dqkv (the q/k/v bias gradients) as ``(dqkv, colsum)``:
    out of the backward kernels where the library supports it, otherwise from a separate pass; final after ``flush()``."""
    mode = L.ATTN_PROXY if size is not None else L.ATTN_CAUSAL
    M, N, Lp = size if size is not None else (0, 1, S)
    dqkv = torch.empty_like(qkv)
    ws = _attn_ws(mode, B, H, M, N, Lp, qkv.device)
    nrows = int(L.lib().xp_attn_bwd_colsum_rows(mode, B, H, S, M, N, Lp, _dt(qkv))) if colsum_defer is not None else 0
    part = colsum_defer.slot(nrows * 3 * H * 64 * 4, colsum_name) if nrows else None
    L.check(L.lib().xp_attn_bwd2(_p(qkv), 3 * H * 64, _p(out), _p(dout), H * 64, _p(stats), _p(pad_mask), _p(dqkv),
                                 float(q_scale), mode, B, H, S, M, N, Lp, _dt(qkv), _p(ws), ws.numel(), _p(part), _stream()),
            "xp_attn_bwd")
    if colsum_defer is None:
        return dqkv
    if not nrows:
        return dqkv, colsum_deferred(dqkv, B * S, 3 * H * 64, colsum_defer, name=colsum_name)
    cs = torch.empty(3 * H * 64, dtype=torch.float32, device=qkv.device)
    colsum_defer.add(part, 0, cs, nrows, 3 * H * 64, 3 * H * 64)
    return dqkv, cs
