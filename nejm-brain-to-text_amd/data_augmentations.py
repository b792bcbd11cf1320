"""gauss_smooth — same 5-argument surface as model_training/data_augmentations.py:6, executed by the
fused HIP stencil (csrc/elementwise.hip): coalesced float4 reads along the feature axis, a 9-tap
sliding window along T in registers, no permute copies."""
import torch

import b2t_ops as ops


def gauss_smooth(inputs, device, smooth_kernel_std=2, smooth_kernel_size=100, padding='same'):
    """inputs [B,T,N] on the HIP device -> smoothed [B,T,N] ('same') or [B,T-K+1,N] ('valid')."""
    x = inputs
    if not x.is_cuda:
        raise RuntimeError("gauss_smooth needs the input on the HIP device; this package has no CPU path")
    if x.dtype != torch.float32:
        x = x.float()
    return ops.augment_smooth(x.contiguous(), smooth_kernel_std, smooth_kernel_size, padding=padding)
