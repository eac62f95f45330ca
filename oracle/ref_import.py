"""Import the UNMODIFIED reference (microsoft/XPretrain, CLIP-ViP) for oracle pinning.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference tree lives at
/root/reference in the build container and does not exist on the GPU box, so this
module is used by ``tests/golden/make_golden.py`` (fixture generation) and by the
CPU tests that cross-check ``oracle.clipvip_oracle`` against the real thing when the
tree is present.

The reference's ``src/modeling/VidCLIP.py`` imports ``src/modeling/CLIP.py`` which
imports ``easydict`` (CLIP.py:20); that package is not installed here, so a
minimal attribute-dict stand-in is registered in ``sys.modules`` first.
"""
import os
import sys
import types
import json
import tempfile

REFERENCE_ROOT = os.environ.get("XPRETRAIN_REFERENCE", "/root/reference")
CLIPVIP_ROOT = os.path.join(REFERENCE_ROOT, "CLIP-ViP")


def available() -> bool:
    return os.path.isfile(os.path.join(CLIPVIP_ROOT, "src", "modeling", "CLIP_ViP.py"))


class _AttrDict(dict):
    """Minimal easydict.EasyDict replacement: nested attribute access."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _AttrDict):
            v = _AttrDict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    __setattr__ = __setitem__


def _install_stubs():
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")
        m.EasyDict = _AttrDict
        sys.modules["easydict"] = m


def load():
    """Returns a namespace with the reference's CLIP_ViP / VidCLIP / loss modules."""
    if not available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_ROOT}")
    _install_stubs()
    if CLIPVIP_ROOT not in sys.path:
        sys.path.insert(0, CLIPVIP_ROOT)
    import importlib
    ns = types.SimpleNamespace()
    ns.CLIP_ViP = importlib.import_module("src.modeling.CLIP_ViP")
    ns.VidCLIP = importlib.import_module("src.modeling.VidCLIP")
    ns.loss = importlib.import_module("src.optimization.loss")
    ns.AttrDict = _AttrDict
    return ns


def make_args(clip_config_dict, add_cls_num=3, temporal_size=12, logit_scale_init_value=4.60,
              if_use_temporal_embed=1):
    """Build the ``args`` object VidCLIP.__init__ expects (VidCLIP.py:9-27) from a
    config dict, writing config.json into a temp dir (no hub access here)."""
    d = tempfile.mkdtemp(prefix="clipvip_cfg_")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(clip_config_dict, f)
    return _AttrDict(
        clip_config=d,
        clip_weights="",
        clip_vision_additional_config=dict(
            type="ViP", temporal_size=temporal_size, if_use_temporal_embed=if_use_temporal_embed,
            logit_scale_init_value=logit_scale_init_value, add_cls_num=add_cls_num),
    )
