"""The benchmark workload of BASELINE.json, stated once for `bench.py` and the tools: model dimensions, synthetic inputs and the
algorithmic FLOP count.  Product-side on purpose: the benchmark must not need test infrastructure (`oracle/`) to build its model;
`tests/test_tools_cpu.py` pins these definitions to the oracle's (same config dict, same tensors from the same seed, same FLOPs).

Reference: the dimensions are openai/clip-vit-base-patch{16,32}'s `config.json`, which `CLIP-ViP/src/modeling/VidCLIP.py:18-23` loads through
`CLIPConfig.from_pretrained`; inputs follow SURVEY.md 8(d) (`[B,T,3,H,W]` fp32 frames, ids with BOS / first-EOT argmax pooling,
`CLIP_ViP.py:880-883`); FLOPs are BASELINE.md 3."""
import torch


def hf_config_dict(vision_hidden, vision_heads, vision_layers, vision_inter, patch, image,
                   text_hidden, text_heads, text_layers, text_inter, vocab, max_pos, proj) -> dict:
    """A CLIPConfig-compatible dict (the shape of openai/clip-vit-base-patch{16,32}'s config.json)."""
    text = {"model_type": "clip_text_model", "hidden_size": text_hidden, "intermediate_size": text_inter, "num_attention_heads": text_heads,
            "num_hidden_layers": text_layers, "max_position_embeddings": max_pos, "vocab_size": vocab, "hidden_act": "quick_gelu",
            "layer_norm_eps": 1e-5, "attention_dropout": 0.0, "projection_dim": proj, "bos_token_id": 0, "eos_token_id": 2, "pad_token_id": 1}
    vision = {"model_type": "clip_vision_model", "hidden_size": vision_hidden, "intermediate_size": vision_inter,
              "num_attention_heads": vision_heads, "num_hidden_layers": vision_layers, "image_size": image, "patch_size": patch,
              "hidden_act": "quick_gelu", "layer_norm_eps": 1e-5, "attention_dropout": 0.0, "projection_dim": proj, "num_channels": 3}
    return {"model_type": "clip", "projection_dim": proj, "logit_scale_init_value": 2.6592, "initializer_factor": 1.0,
            "text_config": text, "vision_config": vision}


def vit_b_config(patch: int = 16, image: int = 224) -> dict:
    """ViT-B/{patch} video tower + the 12-layer / 512-wide CLIP text tower"""
    return hf_config_dict(768, 12, 12, 3072, patch, image, 512, 8, 12, 2048, 49408, 77, 512)


def synthetic_inputs(B, T, R, Lt, vocab=49408, seed=4321, dtype=torch.float32):
    """(video [B,T,3,R,R], ids [B,Lt], mask [B,Lt]) on the CPU: standard-normal frames; ids with BOS = vocab-2 first, EOT = vocab-1
    from a per-row random position >= 2 to the end; mask 1 up to and including the first EOT."""
    g = torch.Generator().manual_seed(seed)
    video = torch.randn(B, T, 3, R, R, generator=g, dtype=dtype)
    bos, eot = vocab - 2, vocab - 1
    ids = torch.randint(1, bos, (B, Lt), generator=g)
    ids[:, 0] = bos
    eot_pos = torch.randint(2, Lt, (B,), generator=g)
    ar = torch.arange(Lt)[None]
    ids = torch.where(ar >= eot_pos[:, None], torch.full_like(ids, eot), ids)
    mask = (ar <= eot_pos[:, None]).long()
    return video, ids, mask


def flops_per_pair(T, R, Lt, patch=16, D=768, Dff=3072, Dt=512, Dfft=2048, Ly=12, add_cls=3, proj=512):
    """(video tower, text tower) algorithmic forward FLOPs of one video-text pair: patch GEMM, per layer the four projections, the
    MLP, frame-window + proxy attention, and the feature projection"""
    L = (R // patch) ** 2
    M = 1 + add_cls
    S = M + T * L
    f_vis = 2 * T * L * (3 * patch * patch) * D + Ly * (8 * S * D * D + 4 * S * D * Dff + 4 * T * L * (M + L) * D + 4 * M * S * D) + 2 * D * proj
    f_txt = Ly * (8 * Lt * Dt * Dt + 4 * Lt * Dt * Dfft + 4 * Lt * Lt * Dt) + 2 * Dt * proj
    return f_vis, f_txt


class ModelArgs:
    """the `args` object VidCLIP's constructor reads (`VidCLIP.py:14-30`): an in-memory config instead of a checkpoint directory"""

    def __init__(self, cfg, temporal_size=12, add_cls_num=3):
        self.clip_config = cfg
        self.clip_weights = ""
        self.clip_vision_additional_config = dict(type="ViP", temporal_size=temporal_size, if_use_temporal_embed=1,
                                                  logit_scale_init_value=4.6, add_cls_num=add_cls_num)
