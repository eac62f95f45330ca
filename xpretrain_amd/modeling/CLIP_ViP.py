"""CLIP-ViP model surface on the MI355X HIP kernels -- drop-in for the reference's
``src/modeling/CLIP_ViP.py`` (same class names, constructor arguments, parameter/buffer names and
``state_dict`` keys -- ``pre_layrnorm`` typo included -- and the same forward keyword names).

The ``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Embedding`` / ``nn.Conv2d`` sub-modules are used purely as
*parameter containers* so that checkpoints round-trip key-for-key; their ``forward`` is
never called.  All arithmetic runs in ``libxpretrain_hip.so`` through ``xpretrain_amd.functional``.
There is no CPU path: tensors must live on an MI355X.

Deliberate differences from the reference (SURVEY.md §8a "known defects"):
  * ``CLIPVisionModel(config, additional_vision_config)`` accepts the ViP config (the reference's
    constructor drops it and raises AttributeError at CLIP_ViP.py:148).
  * half precision is bf16 with fp32 softmax/LayerNorm statistics (the reference is fp16 under apex O2).
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from .. import functional as XF

try:  # config classes only -- same import the reference uses (CLIP_ViP.py:36)
    from transformers.models.clip.configuration_clip import CLIPConfig, CLIPTextConfig, CLIPVisionConfig
except Exception:  # pragma: no cover - transformers is present in the image
    CLIPConfig = CLIPTextConfig = CLIPVisionConfig = None


class _Cfg(dict):
    """attribute-style view of a plain dict (stands in for easydict / HF sub-configs)."""
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def _as_cfg(c):
    if c is None or hasattr(c, "hidden_size") or not isinstance(c, dict):
        return c
    return _Cfg(c)


class _Lazy:
    """a field that is cheap but not free and that the training step never reads: computed on first access"""

    def __init__(self, fn):
        grad = torch.is_grad_enabled()            # evaluate under the grad mode of the forward that created the field,

        def run():                                # not under whatever is current at access time
            with torch.set_grad_enabled(grad):
                return fn()
        self.fn = run


class _Output(dict):
    """dict with attribute access.  Every field of the reference's output class is present: fields the hot path does not
    need are ``_Lazy`` thunks resolved on first access (so ``out.logits_per_text`` or ``out.last_hidden_state`` always
    holds what the reference returns), and an unknown name raises instead of returning None."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if isinstance(v, _Lazy):
            v = v.fn()
            dict.__setitem__(self, k, v)
        return v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(f"{type(self).__name__} has no field {k!r}") from None

    def get(self, k, default=None):
        return self[k] if k in self else default

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def _resolve_all(self):
        for k in list(self.keys()):
            self[k]
        return self

    def copy(self):                    # CPython's dict.copy / dict(out) / pickle bypass __getitem__: resolve first
        return type(self)(**{k: self[k] for k in self.keys()})

    def __reduce__(self):
        return (type(self), (), None, None, iter(self._resolve_all().items()))

    def to_tuple(self):
        return tuple(v for v in self.values() if v is not None)


class CLIPOutput(_Output):
    """loss, logits_per_image, logits_per_text, text_embeds, image_embeds, text_model_output, vision_model_output
    (reference CLIPOutput, CLIP_ViP.py:76-111)."""


class BaseModelOutputWithPooling(_Output):
    def __getitem__(self, k):
        if isinstance(k, int):
            return (self["last_hidden_state"], self["pooler_output"])[k]
        return _Output.__getitem__(self, k)


# ------------------------------------------------------------------------------------------ embeddings
class CLIPVisionViPEmbeddings(nn.Module):
    """reference: CLIP_ViP.py:142-197"""

    def __init__(self, config, additional_vision_config=None):
        super().__init__()
        add = _as_cfg(additional_vision_config)
        if add is None:
            raise ValueError("CLIPVisionViPEmbeddings needs additional_vision_config "
                             "(temporal_size, if_use_temporal_embed, add_cls_num)")
        self.config = config
        self.embed_dim = config.hidden_size
        self.image_size = config.image_size
        self.temporal_size = add.temporal_size
        self.if_use_temporal_embed = add.if_use_temporal_embed
        self.patch_size = config.patch_size
        self.add_cls_num = add.add_cls_num
        self.added_cls = nn.Parameter(torch.randn(self.add_cls_num, self.embed_dim))
        self.class_embedding = nn.Parameter(torch.randn(self.embed_dim))
        self.patch_embedding = nn.Conv2d(3, self.embed_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=False)
        self.num_patches = (self.image_size // self.patch_size) ** 2
        self.num_positions = self.num_patches + 1
        self.position_embedding = nn.Embedding(self.num_positions, self.embed_dim)
        self.register_buffer("position_ids", torch.arange(self.num_positions).expand((1, -1)))
        if self.if_use_temporal_embed:
            self.temporal_embedding = nn.Parameter(torch.zeros(1, self.temporal_size, self.embed_dim))

    def forward(self, pixel_values: torch.Tensor, dtype=torch.bfloat16):
        B, T, C, H, W = pixel_values.shape
        if self.if_use_temporal_embed:
            te = self.temporal_embedding
            if T != te.shape[1]:   # tiny [1,D,Tt] tensor: stays in torch so autograd reaches the parameter (:171-174)
                te = F.interpolate(te.transpose(1, 2), size=T, mode="linear").transpose(1, 2)
            time_table = te[0]
        else:
            time_table = torch.zeros(T, self.embed_dim, device=pixel_values.device)
        pw = self.patch_embedding.weight
        x = XF.VisionEmbedFn.apply(pixel_values, pw, self.class_embedding, self.added_cls, self.position_embedding.weight, time_table, dtype,
                                   torch.is_grad_enabled() and pw.requires_grad)
        M = 1 + self.add_cls_num
        L = (H // self.patch_size) * (W // self.patch_size)
        return x, (M, T, L)          # x: [B*(M+T*L), D]


class CLIPTextEmbeddings(nn.Module):
    """reference: CLIP_ViP.py:199-227"""

    def __init__(self, config):
        super().__init__()
        embed_dim = config.hidden_size
        self.token_embedding = nn.Embedding(config.vocab_size, embed_dim)
        self.position_embedding = nn.Embedding(config.max_position_embeddings, embed_dim)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))

    def forward(self, input_ids, position_ids=None, dtype=torch.bfloat16):
        if position_ids is not None:
            raise NotImplementedError("custom position_ids are not on the CLIP-ViP path (run_pretrain.py never passes them)")
        if input_ids.shape[-1] > self.position_embedding.weight.shape[0]:
            raise ValueError("sequence longer than max_position_embeddings")
        return XF.TextEmbedFn.apply(input_ids, self.token_embedding.weight, self.position_embedding.weight, dtype)


# ------------------------------------------------------------------------------------------ blocks
class CLIPAttention(nn.Module):
    """parameter container for q/k/v/out projections (reference: CLIP_ViP.py:230-381)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.embed_dim // self.num_heads
        assert self.head_dim * self.num_heads == self.embed_dim
        self.scale = self.head_dim ** -0.5
        self.dropout = config.attention_dropout
        if self.dropout:
            raise NotImplementedError("attention_dropout > 0 is not on the CLIP-ViP path (all shipped configs use 0.0)")
        self.k_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.v_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.q_proj = nn.Linear(self.embed_dim, self.embed_dim)
        self.out_proj = nn.Linear(self.embed_dim, self.embed_dim)


class CLIPMLP(nn.Module):
    """reference: CLIP_ViP.py:384-396"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        if config.hidden_act != "quick_gelu":
            raise NotImplementedError(f"hidden_act={config.hidden_act!r}: the fused MLP epilogue implements quick_gelu "
                                      "(every openai/clip-* config)")
        self.fc1 = nn.Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = nn.Linear(config.intermediate_size, config.hidden_size)


class CLIPEncoderLayer(nn.Module):
    """reference: CLIP_ViP.py:399-467"""

    def __init__(self, config):
        super().__init__()
        self.embed_dim = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.self_attn = CLIPAttention(config)
        self.layer_norm1 = nn.LayerNorm(self.embed_dim)
        self.mlp = CLIPMLP(config)
        self.layer_norm2 = nn.LayerNorm(self.embed_dim)

    def forward(self, hidden_states, B, S, inputs_size=None, pad_mask=None, side=None, split=None):
        return XF.encoder_layer(hidden_states, self, B, S, self.num_heads, inputs_size, pad_mask, side, split)


def _has_forward_hooks(layers) -> bool:
    import torch.nn.modules.module as M
    if M._global_forward_hooks or M._global_forward_pre_hooks:
        return True
    return any(l._forward_hooks or l._forward_pre_hooks for l in layers)


class CLIPEncoder(nn.Module):
    """reference: CLIP_ViP.py:614-712"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layers = nn.ModuleList([CLIPEncoderLayer(config) for _ in range(config.num_hidden_layers)])
        self.gradient_checkpointing = False

    def forward(self, x, B, S, inputs_size=None, pad_mask=None, collect=None, side=None, collect_side=None):
        """``side``: the proxy rows of ``x`` in fp32 (video tower, bf16 compute: XF.PROXY_SIDE); returns ``(x, side)`` then.
        ``collect`` / ``collect_side``: lists that receive every layer output (compute dtype, as ``last_hidden_state``) and its
        fp32 side rows (``output_hidden_states``)."""
        ckpt = self.gradient_checkpointing and self.training and torch.is_grad_enabled()
        # video tower: two half-batch chains on two streams (functional.ForwardSplit), joined after the last layer.  Training passes
        # keep every layer's buffers in the autograd graph; forward-only passes (and layers without a node: everything frozen under
        # grad mode) have the split hold them until the join.
        split = None
        if (XF.FWD_SPLIT and XF.LAYER_CALLS and inputs_size is not None and pad_mask is None and not ckpt
                and x.is_cuda and B % 2 == 0 and x.shape[0] % 8 == 0 and x.shape[0] >= XF.FWD_SPLIT_MIN_ROWS
                and not _has_forward_hooks(self.layers)):     # (a hook would read a layer's output before the second chain wrote it)
            keeps = torch.is_grad_enabled() and x.requires_grad      # (else the split holds ~12x the stream's bytes per layer until the join)
            if keeps or 12 * len(self.layers) * x.numel() * x.element_size() <= XF.FWD_SPLIT_HOLD_BYTES:
                split = XF.ForwardSplit(x.device)
        for li, layer in enumerate(self.layers):
            if XF.LATE_WEIGHTS["event"] is not None and x.is_cuda:      # the optimizer's overlapped update of the layers >= K (XF.LATE_WEIGHTS)
                XF.wait_late_weights(li, *((torch.cuda.current_stream(x.device), split.stream) if split is not None else ()))
            if ckpt:
                # reference CLIP_ViP.py:675-690 (torch.utils.checkpoint around every encoder layer): the layer's saved-activation
                # arena is dropped after the forward and rebuilt by re-running the layer when its backward starts -- the same
                # kernels on the same inputs, so the gradients are bit-identical to the non-checkpointed run
                from torch.utils.checkpoint import checkpoint
                if side is None:
                    x = checkpoint(lambda t, _l=layer: _l(t, B, S, inputs_size, pad_mask), x, use_reentrant=False)
                else:
                    x, side = checkpoint(lambda t, sd, _l=layer: _l(t, B, S, inputs_size, pad_mask, sd), x, side, use_reentrant=False)
            elif side is None:
                x = layer(x, B, S, inputs_size, pad_mask, None, split)
            else:
                x, side = layer(x, B, S, inputs_size, pad_mask, side, split)
            if collect is not None:
                collect.append(x)
                if collect_side is not None and side is not None:
                    collect_side.append(side)
        if split is not None:
            split.join()
        return x if side is None else (x, side)


# ------------------------------------------------------------------------------------------ towers
class CLIPTextTransformer(nn.Module):
    """reference: CLIP_ViP.py:715-797"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embeddings = CLIPTextEmbeddings(config)
        self.encoder = CLIPEncoder(config)
        self.final_layer_norm = nn.LayerNorm(config.hidden_size)
        self.compute_dtype = torch.bfloat16

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        if input_ids is None:
            raise ValueError("You have to specify either input_ids")
        if output_attentions:
            raise NotImplementedError("the fused attention kernel never materialises attention weights")
        input_ids = input_ids.view(-1, input_ids.shape[-1]).contiguous()
        B, Lt = input_ids.shape
        D = self.config.hidden_size
        x = self.embeddings(input_ids, position_ids, self.compute_dtype)
        pad = None if attention_mask is None else attention_mask.to(torch.int64).contiguous()
        idx = XF.H.argmax_rows(input_ids)
        ln = self.final_layer_norm
        if XF.PROXY_SIDE and self.compute_dtype == torch.bfloat16:
            # the text tower is 256 rows: its WHOLE residual stream is kept in fp32 beside the bf16 rows the kernels read (the same
            # side-row mechanism as the video tower's proxy tokens, with every row a side row)
            emb = self.embeddings
            side = XF.H.text_embed_fwd(input_ids, emb.token_embedding.weight.detach(), emb.position_embedding.weight.detach(), torch.float32)
            hs, hside = ([x], [side]) if output_hidden_states else (None, None)
            x, side = self.encoder(x, B, Lt, None, pad, hs, side, hside)
            pooled = XF.LayerNormFn.apply(XF.GatherRowsFn.apply(x, idx, B, Lt), ln.weight, ln.bias,
                                          XF.H.gather_rows(side, idx, B, Lt, D), (1, 1, 1))
            last = _Lazy(lambda: XF.LayerNormFn.apply(x, ln.weight, ln.bias, side, (1, 1, 1)).view(B, Lt, D))
        else:
            hs, hside = ([x] if output_hidden_states else None), None
            x = self.encoder(x, B, Lt, None, pad, hs)
            # LayerNorm is row-wise, so pooling the EOT rows first and normalising only those is identical to
            # final_layer_norm followed by the gather (:772-776); the full normalised sequence is only produced
            # on request.
            pooled = XF.LayerNormFn.apply(XF.GatherRowsFn.apply(x, idx, B, Lt), ln.weight, ln.bias)
            last = _Lazy(lambda: XF.LayerNormFn.apply(x, ln.weight, ln.bias).view(B, Lt, D))      # (:772) resolved on access
        out = BaseModelOutputWithPooling(last_hidden_state=last, pooler_output=pooled,
                                         hidden_states=None if hs is None else tuple(h.view(B, Lt, D) for h in hs),
                                         hidden_side_rows=None if hside is None else tuple(h.view(B, Lt, D) for h in hside),
                                         attentions=None)
        return out if return_dict is None or return_dict else out.to_tuple()


class CLIPVisionTransformer(nn.Module):
    """reference: CLIP_ViP.py:848-903"""

    def __init__(self, config, additional_vision_config=None):
        super().__init__()
        self.config = config
        self.embeddings = CLIPVisionViPEmbeddings(config, additional_vision_config)
        self.pre_layrnorm = nn.LayerNorm(config.hidden_size)      # (sic) -- checkpoint key
        self.encoder = CLIPEncoder(config)
        self.post_layernorm = nn.LayerNorm(config.hidden_size)
        self.compute_dtype = torch.bfloat16

    def forward(self, pixel_values=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        if pixel_values is None:
            raise ValueError("You have to specify pixel_values")
        if output_attentions:
            raise NotImplementedError("the fused attention kernel never materialises attention weights")
        B = pixel_values.shape[0]
        D = self.config.hidden_size
        x, size = self.embeddings(pixel_values, self.compute_dtype)
        S, M = size[0] + size[1] * size[2], size[0]
        if XF.PROXY_SIDE and self.compute_dtype == torch.bfloat16:
            # the M proxy tokens of every sample travel through the residual stream in fp32 beside the bf16 rows (4 of 2356 rows:
            # the pooled feature is proxy 0 of the last layer, and rounding its residual stream is half of the bf16 path's error)
            emb = self.embeddings
            side = XF.proxy_side_rows(emb.class_embedding, emb.added_cls, emb.position_embedding.weight, B, M)
            x, side = XF.LayerNormFn.apply(x, self.pre_layrnorm.weight, self.pre_layrnorm.bias, side, (S, M, M), True)
            hs, hside = ([x], [side]) if output_hidden_states else (None, None)
            x, side = self.encoder(x, B, S, size, None, hs, side, hside)
            pooled = XF.LayerNormFn.apply(XF.GatherRowsFn.apply(x, None, B, S), self.post_layernorm.weight,
                                          self.post_layernorm.bias, side, (1, 1, M))
        else:
            x = XF.LayerNormFn.apply(x, self.pre_layrnorm.weight, self.pre_layrnorm.bias)
            hs, hside = ([x] if output_hidden_states else None), None
            x = self.encoder(x, B, S, size, None, hs)
            pooled = XF.LayerNormFn.apply(XF.GatherRowsFn.apply(x, None, B, S), self.post_layernorm.weight,
                                          self.post_layernorm.bias)
        out = BaseModelOutputWithPooling(last_hidden_state=x.view(B, S, D), pooler_output=pooled,
                                         hidden_states=None if hs is None else tuple(h.view(B, S, D) for h in hs),
                                         hidden_side_rows=None if hside is None else tuple(h.view(B, M, D) for h in hside),
                                         attentions=None)
        return out if return_dict is None or return_dict else out.to_tuple()


# ------------------------------------------------------------------------------------------ models
class CLIPPreTrainedModel(nn.Module):
    """Weight init of the reference's ``_init_weights`` (CLIP_ViP.py:481-522) + local checkpoint loading."""

    supports_gradient_checkpointing = True          # reference CLIP_ViP.py:478

    def __init__(self, config):
        super().__init__()
        self.config = config

    def _set_gradient_checkpointing(self, module, value=False):
        """reference CLIP_ViP.py:524-526"""
        if isinstance(module, CLIPEncoder):
            module.gradient_checkpointing = value

    def gradient_checkpointing_enable(self):
        """transformers.PreTrainedModel.gradient_checkpointing_enable: every CLIPEncoder below this module recomputes its layers'
        activations in the backward pass (CLIP_ViP.py:675-690)"""
        for m in self.modules():
            self._set_gradient_checkpointing(m, True)

    def gradient_checkpointing_disable(self):
        for m in self.modules():
            self._set_gradient_checkpointing(m, False)

    def _init_weights(self, module):
        factor = getattr(self.config, "initializer_factor", 1.0)
        if isinstance(module, CLIPTextEmbeddings):
            module.token_embedding.weight.data.normal_(mean=0.0, std=factor * 0.02)
            module.position_embedding.weight.data.normal_(mean=0.0, std=factor * 0.02)
        elif isinstance(module, CLIPVisionViPEmbeddings):
            rng = module.config.initializer_range
            nn.init.normal_(module.class_embedding, mean=0.0, std=module.embed_dim ** -0.5 * factor)
            nn.init.normal_(module.patch_embedding.weight, std=rng * factor)
            nn.init.normal_(module.position_embedding.weight, std=rng * factor)
        elif isinstance(module, CLIPAttention):
            in_std = (module.embed_dim ** -0.5) * ((2 * module.config.num_hidden_layers) ** -0.5) * factor
            out_std = (module.embed_dim ** -0.5) * factor
            for m in (module.q_proj, module.k_proj, module.v_proj):
                nn.init.normal_(m.weight, std=in_std)
            nn.init.normal_(module.out_proj.weight, std=out_std)
        elif isinstance(module, CLIPMLP):
            in_std = (module.config.hidden_size ** -0.5) * ((2 * module.config.num_hidden_layers) ** -0.5) * factor
            fc_std = (2 * module.config.hidden_size) ** -0.5 * factor
            nn.init.normal_(module.fc1.weight, std=fc_std)
            nn.init.normal_(module.fc2.weight, std=in_std)
        elif isinstance(module, CLIPModel):
            nn.init.normal_(module.text_projection.weight, std=module.text_embed_dim ** -0.5 * factor)
            nn.init.normal_(module.visual_projection.weight, std=module.vision_embed_dim ** -0.5 * factor)
        if isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def post_init(self):
        self.apply(self._init_weights)

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.bfloat16, torch.float32):
            raise NotImplementedError("compute dtype must be torch.bfloat16 (the production mode) or torch.float32 (the "
                                      "reference's default precision, pretrain/run_pretrain.py:234-236: exact-arithmetic "
                                      "kernels for parity work, not tuned)")
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
        return self

    @classmethod
    def from_pretrained(cls, path, config=None, **kw):
        """Local-directory loader (no hub access): ``pytorch_model.bin`` / ``model.safetensors`` / a ``.pt``
        state dict, loaded non-strictly like the reference's ``from_pretrained`` of plain CLIP weights into the
        ViP model (new ViP parameters keep their init)."""
        if config is None:
            config = load_clip_config(path)
        model = cls(config, **kw)
        sd = None
        cands = [path] if os.path.isfile(path) else [os.path.join(path, f) for f in
                                                      ("pytorch_model.bin", "model.safetensors", "model.pt")]
        for f in cands:
            if os.path.isfile(f):
                if f.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(f)
                else:
                    sd = torch.load(f, map_location="cpu")
                break
        if sd is None:
            raise FileNotFoundError(f"no weights found under {path!r} (there is no hub access; pass a local directory)")
        own = model.state_dict()
        sd = {k: v for k, v in sd.items() if k in own and own[k].shape == v.shape}
        model.load_state_dict(sd, strict=False)
        return model


def load_clip_config(src):
    """``CLIPConfig`` from a directory/file with config.json, a dict, or an existing config object."""
    if CLIPConfig is not None and isinstance(src, CLIPConfig):
        return src
    if isinstance(src, dict):
        d = src
    else:
        f = src if os.path.isfile(src) else os.path.join(src, "config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"{f} not found: there is no hub access, clip_config must be a local directory "
                                    "containing config.json (e.g. a copy of openai/clip-vit-base-patch16's)")
        with open(f) as fh:
            d = json.load(fh)
    return CLIPConfig(text_config=dict(d.get("text_config") or {}), vision_config=dict(d.get("vision_config") or {}),
                      projection_dim=d.get("projection_dim", 512),
                      logit_scale_init_value=d.get("logit_scale_init_value", 2.6592))


class CLIPTextModel(CLIPPreTrainedModel):
    """reference: CLIP_ViP.py:800-846"""

    def __init__(self, config):
        super().__init__(config)
        self.text_model = CLIPTextTransformer(config)
        self.post_init()

    def get_input_embeddings(self):
        return self.text_model.embeddings.token_embedding

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        return self.text_model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                               output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                               return_dict=return_dict)


class CLIPVisionModel(CLIPPreTrainedModel):
    """reference: CLIP_ViP.py:906-950 (whose constructor cannot build the ViP embeddings -- fixed here by taking
    the additional config, also looked up on ``config.vision_additional_config``)."""
    main_input_name = "pixel_values"

    def __init__(self, config, additional_vision_config=None):
        super().__init__(config)
        if additional_vision_config is None:
            additional_vision_config = getattr(config, "vision_additional_config", None)
        self.vision_model = CLIPVisionTransformer(config, additional_vision_config)
        self.post_init()

    def get_input_embeddings(self):
        return self.vision_model.embeddings.patch_embedding

    def forward(self, pixel_values=None, output_attentions=None, output_hidden_states=None, return_dict=None):
        return self.vision_model(pixel_values=pixel_values, output_attentions=output_attentions,
                                 output_hidden_states=output_hidden_states, return_dict=return_dict)


def contrastive_loss(logits):
    """reference: CLIP_ViP.py:66-67 -- CE(logits, diag).  Tiny fp32 glue used only by ``return_loss=True``
    (VidCLIP always passes False; the training loss is ``optimization.loss.NCELearnableTempLoss``)."""
    return F.cross_entropy(logits, torch.arange(len(logits), device=logits.device))


def clip_loss(similarity):
    """reference: CLIP_ViP.py:70-73"""
    return (contrastive_loss(similarity) + contrastive_loss(similarity.T)) / 2.0


class CLIPModel(CLIPPreTrainedModel):
    """reference: CLIP_ViP.py:953-1172"""

    def __init__(self, config):
        super().__init__(config)
        # parameters the optimizer may still be writing on its own stream (XF.LATE_WEIGHTS): state_dict() readers wait
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: XF.join_late_weights())
        text_config, vision_config = config.text_config, config.vision_config
        additional_vision_config = getattr(config, "vision_additional_config", None)
        self.projection_dim = config.projection_dim
        self.text_embed_dim = text_config.hidden_size
        self.vision_embed_dim = vision_config.hidden_size
        self.text_model = CLIPTextTransformer(text_config)
        self.vision_model = CLIPVisionTransformer(vision_config, additional_vision_config)
        self.visual_projection = nn.Linear(self.vision_embed_dim, self.projection_dim, bias=False)
        self.text_projection = nn.Linear(self.text_embed_dim, self.projection_dim, bias=False)
        self.logit_scale = nn.Parameter(torch.ones([]) * config.logit_scale_init_value)
        self.post_init()

    def get_text_features(self, input_ids=None, attention_mask=None, position_ids=None, output_attentions=None,
                          output_hidden_states=None, return_dict=None, if_norm=None):
        out = self.text_model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                              output_attentions=output_attentions, output_hidden_states=output_hidden_states)
        feats = XF.ProjectionFn.apply(out["pooler_output"], self.text_projection.weight)
        return XF.L2NormFn.apply(feats) if if_norm else feats.float()

    def get_image_features(self, pixel_values=None, output_attentions=None, output_hidden_states=None,
                           return_dict=None, if_norm=None):
        out = self.vision_model(pixel_values=pixel_values, output_attentions=output_attentions,
                                output_hidden_states=output_hidden_states)
        feats = XF.ProjectionFn.apply(out["pooler_output"], self.visual_projection.weight)
        return XF.L2NormFn.apply(feats) if if_norm else feats.float()

    def forward(self, input_ids=None, pixel_values=None, attention_mask=None, position_ids=None, return_loss=None,
                output_attentions=None, output_hidden_states=None, return_dict=None):
        # The text tower is ~250 rows: its ~700 kernels per step are launch-latency bound and occupy a few CUs each.
        # Run it (whole tower incl. its projection) on a side HIP stream so it overlaps the video tower's large
        # GEMMs; autograd replays each backward op on its forward stream, so the text backward overlaps too.
        side = self._text_stream(pixel_values.device) if (self.overlap_text_tower and pixel_values is not None
                                                           and pixel_values.is_cuda) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                text_outputs = self.text_model(input_ids=input_ids, attention_mask=attention_mask,
                                               position_ids=position_ids, output_attentions=output_attentions,
                                               output_hidden_states=output_hidden_states)
                text_embeds = XF.L2NormFn.apply(XF.ProjectionFn.apply(text_outputs["pooler_output"], self.text_projection.weight))
                text_done = torch.cuda.Event()
                text_done.record(side)                          # the join below waits for the TOWER, not for what is queued behind it
                CLIPModel.run_deferred_text_stream_work()       # (e.g. the loader's copy of the NEXT batch: behind the text tower's forward)
            vision_outputs = self.vision_model(pixel_values=pixel_values, output_attentions=output_attentions,
                                               output_hidden_states=output_hidden_states)
            image_embeds = XF.L2NormFn.apply(XF.ProjectionFn.apply(vision_outputs["pooler_output"], self.visual_projection.weight))
            main.wait_event(text_done)
            text_embeds.record_stream(main)
            return self._finish(image_embeds, text_embeds, text_outputs, vision_outputs, return_loss, return_dict,
                                output_hidden_states)
        vision_outputs = self.vision_model(pixel_values=pixel_values, output_attentions=output_attentions,
                                           output_hidden_states=output_hidden_states)
        text_outputs = self.text_model(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                       output_attentions=output_attentions, output_hidden_states=output_hidden_states)
        image_embeds = XF.L2NormFn.apply(XF.ProjectionFn.apply(vision_outputs["pooler_output"], self.visual_projection.weight))
        text_embeds = XF.L2NormFn.apply(XF.ProjectionFn.apply(text_outputs["pooler_output"], self.text_projection.weight))
        return self._finish(image_embeds, text_embeds, text_outputs, vision_outputs, return_loss, return_dict,
                            output_hidden_states)

    overlap_text_tower = True          # class attribute: False runs the text tower on the caller's stream (A/B: DESIGN_HISTORY.md 6.0)
    _text_streams = {}          # one per device, shared by every model instance (and by utils.prefetch.PrefetchLoader(stream="text"))

    @staticmethod
    def shared_text_stream(device):
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        st = CLIPModel._text_streams.get(key)
        if st is None:
            st = CLIPModel._text_streams[key] = torch.cuda.Stream(device=torch.device("cuda", key))
        return st

    def _text_stream(self, device):
        return CLIPModel.shared_text_stream(device)

    # One-shot callables that want to enqueue work on the text tower's stream where it is idle: right behind the tower's forward (the
    # stream then has nothing to do until the loss's backward reaches the text features).  utils.prefetch.PrefetchLoader(stream="text")
    # puts the host->device copy of the next batch there; enqueued at hand-over time instead, the copy would sit in FRONT of the tower.
    _deferred_text_work = []

    @staticmethod
    def defer_to_text_stream(fn):
        CLIPModel._deferred_text_work.append(fn)

    @staticmethod
    def cancel_deferred_text_stream_work(fn):
        if fn in CLIPModel._deferred_text_work:
            CLIPModel._deferred_text_work.remove(fn)

    @staticmethod
    def run_deferred_text_stream_work():
        while CLIPModel._deferred_text_work:
            CLIPModel._deferred_text_work.pop(0)()

    def _finish(self, image_embeds, text_embeds, text_outputs, vision_outputs, return_loss, return_dict, output_hidden_states):
        # CLIP_ViP.py:1151-1158: tiny [B,B] fp32 product; VidCLIP never reads it, so it is resolved on access
        logits_per_text = _Lazy(lambda: XF.SimLogitsFn.apply(text_embeds, image_embeds, self.logit_scale))
        logits_per_image = _Lazy(lambda: XF.SimLogitsFn.apply(text_embeds, image_embeds, self.logit_scale).T)
        loss = None          # the reference returns None as well unless return_loss (:1160-1162)
        if return_loss:      # clip_loss (:70-73) == NCELearnableTempLoss / 2, through the fused HIP loss kernel
            loss = XF.NCELossFn.apply(image_embeds, text_embeds, self.logit_scale) * 0.5
        out = CLIPOutput(loss=loss, logits_per_image=logits_per_image, logits_per_text=logits_per_text,
                         text_embeds=text_embeds, image_embeds=image_embeds, text_model_output=text_outputs,
                         vision_model_output=vision_outputs)
        if return_dict is not None and not return_dict:      # the reference's tuple form (CLIP_ViP.py:1160-1162)
            t = (out.logits_per_image, out.logits_per_text, text_embeds, image_embeds, text_outputs, vision_outputs)
            return ((loss,) + t) if loss is not None else t
        return out
