from .CLIP_ViP import (CLIPModel, CLIPTextModel, CLIPVisionModel, CLIPOutput, CLIPVisionTransformer,  # noqa: F401
                       CLIPTextTransformer, CLIPEncoder, CLIPEncoderLayer, CLIPAttention, CLIPMLP,
                       CLIPVisionViPEmbeddings, CLIPTextEmbeddings, load_clip_config, clip_loss)
from .VidCLIP import VidCLIP  # noqa: F401
