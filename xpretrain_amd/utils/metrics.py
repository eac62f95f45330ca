"""Retrieval metrics on the device (reference: ``src/utils/metrics.py`` and ``validate`` in
``src/tasks/run_video_retrieval.py:123-203``).

Same function names and return tuples as the reference -- ``cal_cossim``, ``np_softmax`` (axis 0 only, the one DSL
uses), ``compute_metrics``, ``compute_metrics_multi`` -- but they take and keep GPU tensors: the similarity GEMM, the
dual-softmax re-rank and the rank counting run as HIP kernels; only two int32 vectors of length n_queries come back to
the host for the final recall / median / mean arithmetic.  There is no CPU path.
"""
import numpy as np
import torch

from .. import _lib as L
from .. import hip_ops as H


def _f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (xpretrain_amd has no CPU path)")
    return t.contiguous().float()


def cal_cossim(feats1: torch.Tensor, feats2: torch.Tensor) -> torch.Tensor:
    """sim[i][j] = feats1[i] . feats2[j]   (metrics.py:3-5)"""
    a, b = _f32(feats1, "feats1"), _f32(feats2, "feats2")
    sim = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L.check(L.lib().xp_sim_matrix(H._p(a), H._p(b), H._p(sim), a.shape[0], b.shape[0], a.shape[1], H._stream()), "xp_sim_matrix")
    return sim


def _col_softmax(sim: torch.Tensor, theta: float, multiply: bool) -> torch.Tensor:
    out = _f32(sim, "sim").clone()
    L.check(L.lib().xp_dsl_rerank(H._p(out), out.shape[0], out.shape[1], float(theta), int(multiply), H._stream()),
            "xp_dsl_rerank")
    return out


def dsl_rerank(sim: torch.Tensor, theta: float = 100.0) -> torch.Tensor:
    """``sim * np_softmax(sim * theta, axis=0)`` (run_video_retrieval.py:170-171); returns a new tensor."""
    return _col_softmax(sim, theta, True)


def np_softmax(X: torch.Tensor, theta: float = 1.0, axis=0) -> torch.Tensor:
    """metrics.py:7-39 for the one case the path uses: softmax over axis 0 of a 2-D matrix."""
    if axis != 0 or X.dim() != 2:
        raise NotImplementedError("np_softmax: only axis=0 of a 2-D matrix (the DSL use) is on the HIP path")
    return _col_softmax(X, theta, False)


def _rank_counts(x: torch.Tensor, labels=None, transpose=False):
    x = _f32(x, "sim")
    n, m = x.shape
    q = m if transpose else n
    lab = None
    if labels is not None:
        lab = torch.as_tensor(labels, dtype=torch.int64, device=x.device).contiguous()
        if lab.numel() != q:
            raise ValueError("labels must have one entry per query")
    greater = torch.empty(q, dtype=torch.int32, device=x.device)
    equal = torch.empty(q, dtype=torch.int32, device=x.device)
    L.check(L.lib().xp_retrieval_ranks(H._p(x), H._p(lab) if lab is not None else None, n, m, int(transpose),
                                       H._p(greater), H._p(equal), H._stream()), "xp_retrieval_ranks")
    return greater.cpu().numpy(), equal.cpu().numpy()


def _summarise(greater, equal):
    # np.where(sorted - label == 0) lists EVERY sorted position that ties the label (metrics.py:45-47)
    ind = np.concatenate([np.arange(g, g + e) for g, e in zip(greater, equal)]) if (equal != 1).any() else greater
    r1 = float(np.sum(ind == 0)) / len(ind)
    r5 = float(np.sum(ind < 5)) / len(ind)
    r10 = float(np.sum(ind < 10)) / len(ind)
    return r1, r5, r10, np.median(ind) + 1, np.mean(ind) + 1


def compute_metrics(x: torch.Tensor):
    """(R@1, R@5, R@10, median rank, mean rank) of the diagonal entries of each ROW (metrics.py:41-53)."""
    return _summarise(*_rank_counts(x))


def compute_metrics_multi(x: torch.Tensor, t2v_labels_list):
    return _summarise(*_rank_counts(x, labels=t2v_labels_list))


def retrieval_metrics(text_feats: torch.Tensor, vis_feats: torch.Tensor):
    """The body of ``validate`` (run_video_retrieval.py:163-188): returns {setting: {direction: metrics}} for the
    'simple' and 'DSL' settings; the v2t direction ranks the columns of the same matrix (no transposed copy)."""
    sim = cal_cossim(text_feats, vis_feats)
    out = {}
    for setting in ("simple", "DSL"):
        if setting == "DSL":
            sim = dsl_rerank(sim, 100.0)
        out[setting] = {"v2t": _summarise(*_rank_counts(sim, transpose=True)), "t2v": _summarise(*_rank_counts(sim))}
    return out
