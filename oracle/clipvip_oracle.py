"""Functional CPU restatement of the CLIP-ViP contrastive hot path (the ORACLE).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Never imported by the product
package ``xpretrain_amd``; the product path has no CPU fallback.

Every function works on a *flat state dict* with the reference's checkpoint key
schema (``VidCLIP.state_dict()`` minus the ``clipmodel.`` prefix) and
plain tensors, computes in whatever dtype the tensors carry (fp32 for parity tests,
fp64 for tight kernel checks), and is differentiable through ``torch.autograd`` so
gradients come for free.  Citations are file:line into
``/root/reference/CLIP-ViP/src``.

Parity pinning: ``tests/golden/make_golden.py`` runs the UNMODIFIED reference
(``oracle.ref_import``) on seeded inputs and stores its outputs/gradients as
fixtures; ``tests/test_oracle_golden.py`` checks this restatement against them
(CPU, every round) and, when /root/reference is present, against the live reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class _RoundBoth(torch.autograd.Function):
    """value AND incoming gradient rounded to the storage dtype (the HIP path keeps the gradient of every bf16 activation in
    bf16 as well: residual-stream, LayerNorm-output, q/k/v, attention-output and MLP gradients)."""

    @staticmethod
    def forward(ctx, t, dtype):
        ctx.dtype = dtype
        return t.to(dtype).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _RoundValueOnly(torch.autograd.Function):
    """value rounded, gradient passed through unrounded"""

    @staticmethod
    def forward(ctx, t, dtype):
        return t.to(dtype).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundGradOnly(torch.autograd.Function):
    """value untouched, incoming gradient rounded"""

    @staticmethod
    def forward(ctx, t, dtype):
        ctx.dtype = dtype
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _Rounding:
    """Optional storage-precision emulation.  With ``ROUND.dtype = torch.bfloat16`` every tensor the HIP path
    stores in bf16 (weights fed to GEMMs, LayerNorm outputs, q/k/v, attention output, softmax probabilities fed
    to P.V, MLP pre-activation/activation, the residual stream) is rounded at the same point here, so a kernel
    can be told apart from bf16 noise: HIP-vs-emulation must agree far tighter than HIP-vs-fp32.  With
    ``ROUND.grads = True`` the gradients of the rounded ACTIVATIONS are rounded too (weights: ``grad=False`` -- parameter
    gradients are fp32 in the HIP path), which makes the emulation a tight reference for the backward.  Default: off."""
    dtype = None
    grads = False
    attn_operands = False      # True: the attention backward in the kernels' flash-attention FORM (row term from the stored output, P / dS
                               # rounded as matrix-core operands: prob / scores / _MaskedCoreStorageEmulation below).  Off for the parity gates:
                               # the oracle's stored output is another bf16 realisation than the kernel's, so mirroring the form adds the
                               # oracle's own row-term noise to the comparison instead of removing the kernel's (measured: text q_proj gradients
                               # 6.7e-2 -> 1.2e-1 at batch 2); tools/grad_scatter_study.py switches it on to measure the distance between the forms
    proxy_fp32 = True          # the HIP path keeps the M proxy rows of the video tower's residual stream in fp32 (functional.PROXY_SIDE)

    def resid(self, t: Tensor, size) -> Tensor:
        """rounding of the RESIDUAL STREAM [B,S,D]: with ``proxy_fp32`` the first M rows of every video-tower sample (``size`` =
        (M, N, L)) and the whole stream of the text tower (``size`` None) stay exact"""
        if self.dtype is None or not self.proxy_fp32:
            return self(t)
        if size is None:
            return t
        M = size[0]
        return torch.cat([t[:, :M], self(t[:, M:])], dim=1)

    def prob(self, w: Tensor) -> Tensor:
        """softmax probabilities as the attention kernels feed them to the matrix cores: P is rounded to the storage dtype for P.V
        (forward) and for dV = P^T.dO (backward, the same rounded P); its gradient dP = dO.V^T stays in fp32 accumulators"""
        if self.dtype is None or not self.attn_operands:
            return w
        return _RoundValueOnly.apply(w, self.dtype) if w.requires_grad else w.to(self.dtype).to(w.dtype)

    def scores(self, s: Tensor) -> Tensor:
        """the attention scores: never stored (fp32 accumulators), but their GRADIENT dS = P o (dP - delta) is the matrix-core
        operand of dQ = dS.K and dK = dS^T.Q and is rounded to the storage dtype there"""
        if self.dtype is None or not self.attn_operands or not (self.grads and s.requires_grad):
            return s
        return _RoundGradOnly.apply(s, self.dtype)

    def __call__(self, t: Tensor, grad: bool = True) -> Tensor:
        if self.dtype is None:
            return t
        if self.grads and grad and t.requires_grad:
            return _RoundBoth.apply(t, self.dtype)
        return t.to(self.dtype).to(t.dtype)


ROUND = _Rounding()
LN_EPS = 1e-5  # nn.LayerNorm default, modeling/CLIP_ViP.py:404-406,855-857


@dataclass
class TowerCfg:
    hidden: int
    heads: int
    layers: int
    intermediate: int


@dataclass
class OracleCfg:
    vision: TowerCfg
    text: TowerCfg
    patch: int
    image: int
    add_cls_num: int
    temporal_size: int
    proj: int
    use_temporal_embed: bool = True

    @staticmethod
    def from_hf_dict(d: dict, add_cls_num=3, temporal_size=12, use_temporal_embed=True) -> "OracleCfg":
        v, t = d["vision_config"], d["text_config"]
        return OracleCfg(
            vision=TowerCfg(v["hidden_size"], v["num_attention_heads"], v["num_hidden_layers"], v["intermediate_size"]),
            text=TowerCfg(t["hidden_size"], t["num_attention_heads"], t["num_hidden_layers"], t["intermediate_size"]),
            patch=v["patch_size"], image=v["image_size"], add_cls_num=add_cls_num,
            temporal_size=temporal_size, proj=d["projection_dim"], use_temporal_embed=use_temporal_embed)


# ----------------------------------------------------------------------------- primitives
def layer_norm(x: Tensor, w: Tensor, b: Tensor, resid_size=None) -> Tensor:
    """nn.LayerNorm over the last dim, biased variance, eps 1e-5 (CLIP_ViP.py:404-406).  ``resid_size``: the output IS the
    residual stream of the video tower (pre_layrnorm): emulated storage rounding follows ROUND.resid."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + LN_EPS) * w + b
    return ROUND.resid(y, resid_size) if resid_size is not None else ROUND(y)


def quick_gelu(x: Tensor) -> Tensor:
    """ACT2FN["quick_gelu"] = x * sigmoid(1.702 x)  (CLIP_ViP.py:388,394)."""
    return x * torch.sigmoid(1.702 * x)


def linear(x: Tensor, sd: Dict[str, Tensor], name: str, bias: bool = True) -> Tensor:
    y = x @ ROUND(sd[name + ".weight"], grad=False).t()
    if bias:
        y = y + sd[name + ".bias"]
    return y


def _heads(t: Tensor, h: int) -> Tensor:
    """[B,S,D] -> [B,h,S,dh]  (CLIPAttention._shape, CLIP_ViP.py:250-251)."""
    B, S, D = t.shape
    return t.view(B, S, h, D // h).transpose(1, 2)


# ----------------------------------------------------------------------------- attention
def proxy_attention_core(q: Tensor, k: Tensor, v: Tensor, size: Tuple[int, int, int]) -> Tensor:
    """Video-proxy attention on head-split tensors [B,h,S,dh] with S = M + N*L.

    Follows CLIPAttention.forward2 (CLIP_ViP.py:332-381): frame-n queries see the M
    proxy keys plus the L keys of frame n (:352-363); the M proxy queries see every key
    (:366-375); outputs are ordered [proxies, frames] (:377).  ``q`` is already scaled.
    """
    M, N, L = size
    B, h, S, dh = q.shape
    qf = q[:, :, M:].reshape(B, h, N, L, dh)
    kf = k[:, :, M:].reshape(B, h, N, L, dh)
    vf = v[:, :, M:].reshape(B, h, N, L, dh)
    kp = k[:, :, :M].unsqueeze(2).expand(B, h, N, M, dh)
    vp = v[:, :, :M].unsqueeze(2).expand(B, h, N, M, dh)
    kk = torch.cat([kp, kf], dim=3)                      # [B,h,N,M+L,dh]
    vv = torch.cat([vp, vf], dim=3)
    w = ROUND.prob(torch.softmax(ROUND.scores(qf @ kk.transpose(-1, -2)), dim=-1))  # [B,h,N,L,M+L]
    of = (w @ vv).reshape(B, h, N * L, dh)
    wp = ROUND.prob(torch.softmax(ROUND.scores(q[:, :, :M] @ k.transpose(-1, -2)), dim=-1))  # [B,h,M,S]
    op = wp @ v
    return ROUND(torch.cat([op, of], dim=2))


def proxy_attention_core_masked(q: Tensor, k: Tensor, v: Tensor, size: Tuple[int, int, int]) -> Tensor:
    """Same function written as dense attention under a block mask:
    proxy rows see all keys; every row sees the proxy keys; frame-n rows see frame-n keys.
    Second, independent statement used to cross-check ``proxy_attention_core``."""
    M, N, L = size
    S = q.shape[2]
    frame = torch.full((S,), -1, dtype=torch.long)
    frame[M:] = torch.arange(N).repeat_interleave(L)
    allow = (frame[:, None] == frame[None, :]) | (frame[None, :] < 0) | (frame[:, None] < 0)
    s = q @ k.transpose(-1, -2)
    s = s.masked_fill(~allow.to(s.device), float("-inf"))
    return torch.softmax(s, dim=-1) @ v


class _MaskedCoreStorageEmulation(torch.autograd.Function):
    """softmax(q k^T + mask) v with the BACKWARD written the way the attention kernels compute it (flash-attention form), for the
    storage-precision emulation only (ROUND.dtype set and ROUND.grads): the probabilities are recomputed, the row term is
    delta = rowsum(dO o O) with the STORED (rounded) output O -- autograd's softmax backward uses sum_j P dP, i.e. the unrounded
    output; the two differ by dO . (O_stored - O), which does not cancel over the keys of a row -- and P / dS enter the matrix
    products rounded.  Forward values are those of the plain expression."""

    @staticmethod
    def forward(ctx, q, k, v, mask, dtype):
        s = q @ k.transpose(-1, -2) + mask
        p = torch.softmax(s, dim=-1)
        o = (p.to(dtype).to(p.dtype) @ v).to(dtype).to(p.dtype)
        ctx.save_for_backward(q, k, v, mask, o)
        ctx.dtype = dtype
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, mask, o = ctx.saved_tensors
        rnd = lambda t: t.to(ctx.dtype).to(t.dtype)
        do = rnd(do)
        p = torch.softmax(q @ k.transpose(-1, -2) + mask, dim=-1)
        dp = do @ v.transpose(-1, -2)
        delta = (do * o).sum(-1, keepdim=True)
        ds = rnd(p * (dp - delta))
        return ds @ k, ds.transpose(-1, -2) @ q, rnd(p).transpose(-1, -2) @ do, None, None


def masked_attention_core(q: Tensor, k: Tensor, v: Tensor, pad_mask: Optional[Tensor]) -> Tensor:
    """Text-tower attention on [B,h,S,dh]: additive -inf causal mask (CLIP_ViP.py:788-797)
    plus additive finfo.min padding mask (_expand_mask, :50-61), both added to the scores
    before softmax (:286-304)."""
    B, h, S, dh = q.shape
    add = torch.full((S, S), float("-inf"), dtype=q.dtype, device=q.device).triu(1)
    if pad_mask is not None:
        inv = 1.0 - pad_mask.to(q.dtype)[:, None, None, :]
        add = add + inv.masked_fill(inv.bool(), torch.finfo(q.dtype).min)
    if ROUND.dtype is not None and ROUND.grads and ROUND.attn_operands and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _MaskedCoreStorageEmulation.apply(q, k, v, add.expand(B, 1, S, S) if add.dim() == 4 else add, ROUND.dtype)
    s = ROUND.scores(q @ k.transpose(-1, -2)) + add
    return ROUND(ROUND.prob(torch.softmax(s, dim=-1)) @ v)


def attention_block(x: Tensor, sd: Dict[str, Tensor], pfx: str, heads: int,
                    size: Optional[Tuple[int, int, int]], pad_mask: Optional[Tensor]) -> Tensor:
    """q/k/v projections (q scaled by dh^-0.5 AFTER the bias, CLIP_ViP.py:341 / :270),
    core attention, head merge, out_proj (:379 / :328)."""
    B, S, D = x.shape
    dh = D // heads
    q = _heads(ROUND(linear(x, sd, pfx + "q_proj") * dh ** -0.5), heads)
    k = _heads(ROUND(linear(x, sd, pfx + "k_proj")), heads)
    v = _heads(ROUND(linear(x, sd, pfx + "v_proj")), heads)
    if size is not None:
        o = proxy_attention_core(q, k, v, size)
    else:
        o = masked_attention_core(q, k, v, pad_mask)
    o = o.transpose(1, 2).reshape(B, S, D)
    return linear(o, sd, pfx + "out_proj")


# ----------------------------------------------------------------------------- encoder
def encoder_layer(x: Tensor, sd: Dict[str, Tensor], pfx: str, heads: int,
                  size: Optional[Tuple[int, int, int]], pad_mask: Optional[Tensor]) -> Tensor:
    """Pre-LN block, CLIPEncoderLayer.forward else-branch (CLIP_ViP.py:444-460)."""
    h = layer_norm(x, sd[pfx + "layer_norm1.weight"], sd[pfx + "layer_norm1.bias"])
    x = ROUND.resid(x + attention_block(h, sd, pfx + "self_attn.", heads, size, pad_mask), size)
    h = layer_norm(x, sd[pfx + "layer_norm2.weight"], sd[pfx + "layer_norm2.bias"])
    h = linear(ROUND(quick_gelu(linear(h, sd, pfx + "mlp.fc1"))), sd, pfx + "mlp.fc2")   # CLIPMLP :392-396
    return ROUND.resid(x + h, size)


def encoder(x: Tensor, sd: Dict[str, Tensor], pfx: str, cfg: TowerCfg,
            size: Optional[Tuple[int, int, int]], pad_mask: Optional[Tensor], collect=None) -> Tensor:
    """CLIPEncoder.forward layer loop (CLIP_ViP.py:673-704)."""
    for i in range(cfg.layers):
        x = encoder_layer(x, sd, f"{pfx}layers.{i}.", cfg.heads, size, pad_mask)
        if collect is not None:
            collect.append(x)
    return x


# ----------------------------------------------------------------------------- vision tower
def temporal_table(sd: Dict[str, Tensor], T: int) -> Tensor:
    """[T,D] temporal embedding; linear interpolation (align_corners=False) when
    T != temporal_size (CLIPVisionViPEmbeddings.forward, CLIP_ViP.py:170-176)."""
    te = sd["vision_model.embeddings.temporal_embedding"]            # [1,Tt,D]
    if T != te.shape[1]:
        te = F.interpolate(te.transpose(1, 2), size=T, mode="linear").transpose(1, 2)
    return te[0]


def vip_embeddings(video: Tensor, sd: Dict[str, Tensor], cfg: OracleCfg):
    """CLIPVisionViPEmbeddings.forward (CLIP_ViP.py:168-197): non-overlapping conv patch
    embed (no bias) == unfold + GEMM; + temporal[t] + position[1+l]; proxies =
    class_embedding / added_cls, all + position[0], no temporal term; concat."""
    B, T, C, H, W = video.shape
    P = cfg.patch
    gh, gw = H // P, W // P
    wconv = sd["vision_model.embeddings.patch_embedding.weight"]      # [D,3,P,P]
    D = wconv.shape[0]
    patches = ROUND(video.reshape(B, T, C, gh, P, gw, P).permute(0, 1, 3, 5, 2, 4, 6).reshape(B, T, gh * gw, C * P * P))
    pe = patches @ ROUND(wconv.reshape(D, -1), grad=False).t()                    # [B,T,L,D]
    pos = sd["vision_model.embeddings.position_embedding.weight"]     # [1+L,D]
    if cfg.use_temporal_embed:
        pe = pe + temporal_table(sd, T)[None, :, None, :]
    pe = pe + pos[1:][None, None]
    cls = sd["vision_model.embeddings.class_embedding"][None, None, :].expand(B, 1, D) + pos[0]
    add = sd["vision_model.embeddings.added_cls"][None].expand(B, -1, D) + pos[0]
    M = 1 + add.shape[1]
    x = ROUND.resid(torch.cat([cls, add, pe.reshape(B, T * gh * gw, D)], dim=1), (M, T, gh * gw))
    return x, (M, T, gh * gw)


def vision_tower(video: Tensor, sd: Dict[str, Tensor], cfg: OracleCfg, collect=None):
    """CLIPVisionTransformer.forward (CLIP_ViP.py:861-903): embeddings -> pre_layrnorm (sic)
    -> encoder -> token 0 -> post_layernorm.  Returns (last_hidden, pooled)."""
    x, size = vip_embeddings(video, sd, cfg)
    if collect is not None:
        collect.append(x)
    x = layer_norm(x, sd["vision_model.pre_layrnorm.weight"], sd["vision_model.pre_layrnorm.bias"], resid_size=size)
    if collect is not None:
        collect.append(x)
    x = encoder(x, sd, "vision_model.encoder.", cfg.vision, size, None, collect)
    pooled = layer_norm(x[:, 0], sd["vision_model.post_layernorm.weight"], sd["vision_model.post_layernorm.bias"])
    return x, pooled


# ----------------------------------------------------------------------------- text tower
def text_tower(ids: Tensor, mask: Optional[Tensor], sd: Dict[str, Tensor], cfg: OracleCfg, collect=None):
    """CLIPTextTransformer.forward (CLIP_ViP.py:726-786): token+position gather (:210-227),
    causal+padding masks, encoder, final_layer_norm, pooled = hidden at ids.argmax(-1)."""
    B, Lt = ids.shape
    x = ROUND.resid(sd["text_model.embeddings.token_embedding.weight"][ids]
                    + sd["text_model.embeddings.position_embedding.weight"][:Lt][None], None)
    if collect is not None:
        collect.append(x)
    x = encoder(x, sd, "text_model.encoder.", cfg.text, None, mask, collect)
    x = layer_norm(x, sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"])
    pooled = x[torch.arange(B), ids.argmax(dim=-1)]
    return x, pooled


# ----------------------------------------------------------------------------- heads + loss
def l2_normalize(x: Tensor) -> Tensor:
    """x / ||x||_2 over the last dim, no epsilon (CLIP_ViP.py:1148-1149)."""
    return x / x.norm(dim=-1, keepdim=True)


def clip_features(video: Tensor, ids: Tensor, mask: Optional[Tensor], sd: Dict[str, Tensor], cfg: OracleCfg):
    """VidCLIP.forward / CLIPModel.forward with return_loss=False (VidCLIP.py:44-52,
    CLIP_ViP.py:1125-1149).  Returns (vis_features, text_features), unit-norm [B,proj]."""
    _, vp = vision_tower(video, sd, cfg)
    _, tp = text_tower(ids, mask, sd, cfg)
    vis = l2_normalize(ROUND(vp @ ROUND(sd["visual_projection.weight"], grad=False).t()))
    txt = l2_normalize(ROUND(tp @ ROUND(sd["text_projection.weight"], grad=False).t()))
    return vis, txt


def nce_learnable_temp_loss(vis: Tensor, txt: Tensor, log_scale: Tensor) -> Tensor:
    """NCELearnableTempLoss.forward (optimization/loss.py:134-141): logits =
    exp(log_scale) * vis @ txt^T ; CE(rows, diag) + CE(cols, diag)  (sum, no 1/2)."""
    logits = vis @ txt.t() * log_scale.exp()
    lbl = torch.arange(logits.shape[0], device=logits.device)
    return F.cross_entropy(logits, lbl) + F.cross_entropy(logits.t(), lbl)


def nce_vsc_fc_loss(vis: Tensor, txt: Tensor, img: Tensor, cap: Tensor, log_scale: Tensor) -> Tensor:
    """NCELearnableTempLoss_vsc_fc.forward (optimization/loss.py:296-324), the pre-training
    default (pretrain_vip_base_16.json:75): six CE terms over video-subtitle, video-caption
    (with the off-diagonal negatives of both re-packed, :307-314) and frame-caption."""
    s = log_scale.exp()
    n = vis.shape[0]
    lbl = torch.arange(n, device=vis.device)
    v2t = vis @ txt.t() * s
    v2c = vis @ cap.t() * s
    f2c = img @ cap.t() * s
    off = ~torch.eye(n, dtype=torch.bool, device=vis.device)
    neg_t = v2t[off].reshape(n, n - 1)
    neg_c = v2c[off].reshape(n, n - 1)
    a = torch.cat([v2t.diagonal()[:, None], neg_t, neg_c], dim=1)
    b = torch.cat([v2c.diagonal()[:, None], neg_t, neg_c], dim=1)
    zero = torch.zeros(n, dtype=torch.long, device=vis.device)
    return (F.cross_entropy(v2t.t(), lbl) + F.cross_entropy(v2c.t(), lbl)
            + F.cross_entropy(a, zero) + F.cross_entropy(b, zero)
            + F.cross_entropy(f2c.t(), lbl) + F.cross_entropy(f2c, lbl))


def full_step(video, ids, mask, sd, cfg: OracleCfg, logit_scale_key="logit_scale"):
    """forward + NCELearnableTempLoss on the local batch; returns (loss, vis, txt)."""
    vis, txt = clip_features(video, ids, mask, sd, cfg)
    return nce_learnable_temp_loss(vis, txt, sd[logit_scale_key]), vis, txt


# ----------------------------------------------------------------------------- helpers
def strip_prefix(sd: Dict[str, Tensor], prefix: str = "clipmodel.") -> Dict[str, Tensor]:
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def hf_config_dict(vision_hidden, vision_heads, vision_layers, vision_inter, patch, image,
                   text_hidden, text_heads, text_layers, text_inter, vocab, max_pos, proj) -> dict:
    """A CLIPConfig-compatible dict (the shape of openai/clip-vit-base-patch{16,32}'s config.json)."""
    return {
        "model_type": "clip", "projection_dim": proj, "logit_scale_init_value": 2.6592,
        "initializer_factor": 1.0,
        "text_config": {"model_type": "clip_text_model", "hidden_size": text_hidden, "intermediate_size": text_inter,
                        "num_attention_heads": text_heads, "num_hidden_layers": text_layers,
                        "max_position_embeddings": max_pos, "vocab_size": vocab, "hidden_act": "quick_gelu",
                        "layer_norm_eps": 1e-5, "attention_dropout": 0.0, "projection_dim": proj,
                        "bos_token_id": 0, "eos_token_id": 2, "pad_token_id": 1},
        "vision_config": {"model_type": "clip_vision_model", "hidden_size": vision_hidden,
                          "intermediate_size": vision_inter, "num_attention_heads": vision_heads,
                          "num_hidden_layers": vision_layers, "image_size": image, "patch_size": patch,
                          "hidden_act": "quick_gelu", "layer_norm_eps": 1e-5, "attention_dropout": 0.0,
                          "projection_dim": proj, "num_channels": 3},
    }


def vit_b_config(patch: int = 16, image: int = 224) -> dict:
    """openai/clip-vit-base-patch{16,32} dimensions."""
    return hf_config_dict(768, 12, 12, 3072, patch, image, 512, 8, 12, 2048, 49408, 77, 512)


def synthetic_inputs(B, T, R, Lt, vocab=49408, seed=4321, dtype=torch.float32):
    """SURVEY.md §8d synthetic inputs: randn frames; ids with BOS=vocab-2, EOT=vocab-1 at a
    per-row random position >=2 then EOT padding; mask 1 up to and including the first EOT."""
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
    """BASELINE.md §3 algorithmic forward FLOPs per video-text pair."""
    L = (R // patch) ** 2
    M = 1 + add_cls
    S = M + T * L
    f_vis = 2 * T * L * (3 * patch * patch) * D + Ly * (8 * S * D * D + 4 * S * D * Dff
                                                          + 4 * T * L * (M + L) * D + 4 * M * S * D) + 2 * D * proj
    f_txt = Ly * (8 * Lt * Dt * Dt + 4 * Lt * Dt * Dfft + 4 * Lt * Lt * Dt) + 2 * Dt * proj
    return f_vis, f_txt
