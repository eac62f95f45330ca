"""``VidCLIP`` -- the drop-in boundary of the hot path (reference: src/modeling/VidCLIP.py:8-103).

Same constructor argument object (``args.clip_config``, ``args.clip_weights``,
``args.clip_vision_additional_config.{type,temporal_size,if_use_temporal_embed,logit_scale_init_value,
add_cls_num}``), same ``forward`` keyword names (the training loop calls ``model(**batch)``,
run_pretrain.py:342), same output dict keys, same ``state_dict`` keys (prefix ``clipmodel.``).
"""
from __future__ import annotations

import torch
from torch import nn

from .CLIP_ViP import CLIPModel, _as_cfg, load_clip_config


class VidCLIP(nn.Module):
    def __init__(self, args):
        super().__init__()
        clipconfig = load_clip_config(args.clip_config)
        add = _as_cfg(args.clip_vision_additional_config)
        setattr(clipconfig, "vision_additional_config", add)
        self.vision_additional_config = add
        if add.type != "ViP":
            raise NotImplementedError(f"clip_vision_additional_config.type={add.type!r}: only the 'ViP' path (every "
                                      "shipped CLIP-ViP config) is built; the 'ST' ablation (modeling/CLIP.py) is out of scope")
        if getattr(args, "clip_weights", None):
            self.clipmodel = CLIPModel.from_pretrained(args.clip_weights, config=clipconfig)
        else:
            self.clipmodel = CLIPModel(clipconfig)
        # init logit scale from the additional config (VidCLIP.py:25-27); the parameter holds the LOG scale
        self.clipmodel.logit_scale.data.fill_(add.logit_scale_init_value)

    def overload_logit_scale(self, overload_logit_scale):
        self.clipmodel.logit_scale.data.fill_(overload_logit_scale)

    def forward(self, video, text_input_ids, text_input_mask, image=None, caption_ids=None, caption_masks=None):
        """video [B, n_clips*num_frms, C, H, W]; text_input_ids/text_input_mask [B, L];
        image [B, img_num, C, H, W]; caption_ids/caption_masks [B, img_num, L]   (VidCLIP.py:32-81)"""
        outputs = self.clipmodel(input_ids=text_input_ids, attention_mask=text_input_mask, pixel_values=video,
                                 return_loss=False)
        results = {"text_features": outputs["text_embeds"], "vis_features": outputs["image_embeds"]}
        if image is not None:     # second pass: middle frame(s) as T=1 "videos" + generated captions (:70-79)
            B, img_num, C, H, W = image.shape
            Lc = caption_ids.shape[-1]
            outputs = self.clipmodel(input_ids=caption_ids.reshape(-1, Lc), attention_mask=caption_masks.reshape(-1, Lc),
                                     pixel_values=image.reshape(-1, 1, C, H, W), return_loss=False)
            results["img_features"] = outputs["image_embeds"]
            results["cap_features"] = outputs["text_embeds"]
        return results

    def forward_video(self, video):
        return self.clipmodel.get_image_features(pixel_values=video, if_norm=True)

    def forward_text(self, text_input_ids, text_input_mask):
        return self.clipmodel.get_text_features(input_ids=text_input_ids, attention_mask=text_input_mask, if_norm=True)

    def freeze_text_encoder(self, freeze_text_proj):
        freeze_list = [self.clipmodel.text_model]
        if freeze_text_proj:
            freeze_list.append(self.clipmodel.text_projection)
        for m in freeze_list:
            m.eval()
            for param in m.parameters():
                param.requires_grad = False
