"""GPU: retrieval evaluation and uint8 frame ingest (8f-4) against the reference-generated fixture."""
import numpy as np
import pytest
import torch

from oracle import clipvip_oracle as O
from oracle import retrieval_oracle as RO
from tests.gpu_util import report

pytestmark = pytest.mark.gpu


def test_metrics_against_reference_fixture(golden):
    from xpretrain_amd.utils import metrics
    fx = golden("retrieval.pt")
    for ci, c in enumerate(fx["cases"]):
        txt, vis = c["txt"].cuda(), c["vis"].cuda()
        got = metrics.retrieval_metrics(txt, vis)
        n = txt.shape[0]
        for setting in ("simple", "DSL"):
            for direction in ("v2t", "t2v"):
                g, r = got[setting][direction], c["results"][setting][direction]
                # ranks are integers: 'simple' must agree exactly; after the fp32 dual softmax a near-tie may swap
                tol = 0.0 if setting == "simple" else 1.5 / n
                assert all(abs(a - b) <= tol * max(1.0, abs(b)) + (0 if setting == "simple" else 0.51) * (i >= 3)
                           for i, (a, b) in enumerate(zip(g, r))), (ci, setting, direction, g, r)
        sim = metrics.cal_cossim(txt, vis)
        assert metrics.compute_metrics_multi(sim, c["labels"].tolist()) == pytest.approx(c["multi"], rel=1e-12)
        assert metrics.compute_metrics(sim) == pytest.approx(c["results"]["simple"]["t2v"], rel=1e-12)
        if c["sim"] is not None:
            assert report(f"cal_cossim case {ci}", sim, c["sim"], 1e-5) <= 1e-5
            assert report(f"np_softmax(100x) case {ci}", metrics.np_softmax(sim * 100.0, 1.0, axis=0), c["softmax100"], 1e-4) <= 1e-4


def test_retrieval_eval_through_the_model():
    """validate()'s loop shape: inference forward at T=12 with 50-token captions, features -> metrics; the metrics of the
    HIP features equal those of the oracle features (tiny model, 24 pairs whose captions are tied to the videos only
    through random weights, so ranks are generic)."""
    from xpretrain_amd.modeling import VidCLIP
    from xpretrain_amd.utils import metrics
    from tests.test_model_gpu import _Args
    torch.manual_seed(3)
    cfgd = O.hf_config_dict(128, 2, 2, 256, 16, 32, 128, 2, 2, 256, 120, 64, 64)
    model = VidCLIP(_Args(cfgd, 12)).cuda().eval()
    with torch.no_grad():
        model.clipmodel.vision_model.embeddings.temporal_embedding.normal_(0, 0.1)
    video, ids, mask = O.synthetic_inputs(24, 12, 32, 50, vocab=120)
    with torch.no_grad():
        feats = [model(video[i:i + 8].cuda(), ids[i:i + 8].cuda(), mask[i:i + 8].cuda()) for i in range(0, 24, 8)]
    vis = torch.cat([f["vis_features"] for f in feats]); txt = torch.cat([f["text_features"] for f in feats])
    sd = {k: v.detach().cpu() for k, v in O.strip_prefix(model.state_dict()).items()}
    rv, rt = O.clip_features(video, ids, mask, sd, O.OracleCfg.from_hf_dict(cfgd))
    assert (vis.cpu() - rv).abs().max() < 2e-2 and (txt.cpu() - rt).abs().max() < 2e-2
    got = metrics.retrieval_metrics(txt, vis)
    ref = RO.validate(txt.cpu().numpy(), vis.cpu().numpy())          # same features: kernel vs numpy
    for setting in ("simple", "DSL"):
        for direction in ("v2t", "t2v"):
            assert got[setting][direction] == pytest.approx(ref[setting][direction], rel=1e-6, abs=0.6 if setting == "DSL" else 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_uint8_ingest_matches_normalised_float_path(golden, dtype):
    """xp_im2col_u8 == ImageNorm arithmetic (fixture) followed by the float im2col."""
    from xpretrain_amd import hip_ops as H
    fx = golden("retrieval.pt")
    frames = fx["frames"].reshape(-1, 3, 32, 32).cuda()
    normed = fx["normed"].reshape(-1, 3, 32, 32).cuda()
    a = H.im2col_u8(frames, 16, dtype)
    b = H.im2col(normed.contiguous(), 16, dtype)
    tol = 1e-6 if dtype == torch.float32 else 4e-3          # bf16: at most 1 ulp from a differently rounded fp32 value
    assert report(f"im2col_u8 {dtype}", a, b, tol) <= tol


def test_uint8_video_through_the_model():
    from xpretrain_amd.modeling import VidCLIP
    from tests.test_model_gpu import _Args
    torch.manual_seed(4)
    cfgd = O.hf_config_dict(128, 2, 1, 256, 16, 32, 128, 2, 1, 256, 120, 16, 64)
    model = VidCLIP(_Args(cfgd, 2)).cuda().eval()
    _, ids, mask = O.synthetic_inputs(2, 2, 32, 8, vocab=120)
    frames = torch.randint(0, 256, (2, 2, 3, 32, 32), dtype=torch.uint8)
    ref = torch.from_numpy(RO.image_norm(frames.numpy(), H_mean(), H_std()))
    with torch.no_grad():
        a = model(frames.cuda(), ids.cuda(), mask.cuda())["vis_features"]
        b = model(ref.cuda(), ids.cuda(), mask.cuda())["vis_features"]
    assert (a - b).abs().max().item() < 5e-3


def H_mean():
    from xpretrain_amd import hip_ops as H
    return H.CLIP_MEAN


def H_std():
    from xpretrain_amd import hip_ops as H
    return H.CLIP_STD
