"""CPU restatement of the retrieval evaluation (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Restates ``src/utils/metrics.py`` (``cal_cossim`` :3-5, ``np_softmax`` :7-39, ``compute_metrics`` :41-53,
``compute_metrics_multi`` :55-69) and the simple / DSL evaluation of ``validate``
(``src/tasks/run_video_retrieval.py:163-188``) in numpy, and the device-side frame normalisation of
``ImageNorm`` (``src/datasets/data_utils.py:256-281``).  Pinned by tests/golden/retrieval.pt, which
tests/golden/make_golden.py::retrieval generates by calling the reference's own functions.
"""
import numpy as np


def cal_cossim(a, b):
    return np.dot(a, b.T)


def col_softmax(x, theta):
    y = x * float(theta)
    y = y - np.max(y, axis=0, keepdims=True)
    y = np.exp(y)
    return y / np.sum(y, axis=0, keepdims=True)


def ranks(x, labels=None):
    """positions (0-based, descending order) of each row's labelled entry; ties list every tying position"""
    sx = np.sort(-x, axis=1)
    lab = np.arange(x.shape[0]) if labels is None else np.asarray(labels)
    d = -x[np.arange(x.shape[0]), lab][:, None]
    return np.where(sx - d == 0)[1]


def summarise(ind):
    return (float(np.sum(ind == 0)) / len(ind), float(np.sum(ind < 5)) / len(ind), float(np.sum(ind < 10)) / len(ind),
            np.median(ind) + 1, np.mean(ind) + 1)


def validate(text_feats, vis_feats):
    sim = cal_cossim(text_feats, vis_feats)
    out = {}
    for setting in ("simple", "DSL"):
        if setting == "DSL":
            sim = sim * col_softmax(sim, 100)
        out[setting] = {"v2t": summarise(ranks(sim.T)), "t2v": summarise(ranks(sim))}
    return out


def image_norm(frames_u8, mean, std):
    """uint8 [...,3,H,W] -> float32: /255, -mean, /std per channel (ImageNorm.__call__, data_utils.py:271-281)."""
    x = frames_u8.astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32).reshape(3, 1, 1)
    s = np.asarray(std, np.float32).reshape(3, 1, 1)
    return (x - m) / s
