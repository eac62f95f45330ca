"""xpretrain_amd -- MI355X-native (gfx950) implementation of microsoft/XPretrain's CLIP-ViP
video-text contrastive hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed/RCCL only).
Arithmetic: hand-written HIP kernels in ``libxpretrain_hip.so`` reached through the C ABI
declared in ``include/xpretrain_hip.h``.  There is no CPU or eager-PyTorch fallback: importing
the ops without the built library raises.
"""
__version__ = "0.1.0"
