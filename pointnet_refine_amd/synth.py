"""Synthetic lane-segment batches generated ON DEVICE: the
benchmark and the examples use this so that the host dataloader is out of the picture.
Distribution follows the reference's data contract: +-25 m slices
(tools/generate_train_data.py:11,260), tube of context points around the line, everything
centred on the noisy line's centroid (src/dataset.py:229-235), raw un-normalised intensity in channel 3
(src/dataset.py:65-68,234), noisy 32-point polylines (tools/augment_train_data.py:23-75)."""
from __future__ import annotations

import math

import torch


def synthetic_batch(B: int, N: int, device, seed: int = 1234, M: int = 32, C: int = 4):
    """(context (B,N,C), noisy_line (B,M,3), target_offset (B,M,3)) as LaneRefineDataset.__getitem__
    hands them over (src/dataset.py:205-253): context points scattered in a tube around the TRUE
    line, line and context both expressed relative to the NOISY line's centroid (:229-235) - so
    the offset is observable from the context and the task is learnable, as with real data."""
    g = torch.Generator(device=device).manual_seed(seed)
    f32 = dict(device=device, dtype=torch.float32)
    xs = torch.linspace(-25, 25, M, **f32)
    a = torch.randn(B, 1, generator=g, **f32) * 0.05          # true line: y = a + b x, z = 0
    b = torch.randn(B, 1, generator=g, **f32) * 0.002
    gt = torch.zeros(B, M, 3, **f32)
    gt[..., 0] = xs
    gt[..., 1] = a + b * xs
    ctx = torch.empty(B, N, C, **f32)
    ctx[..., 0] = torch.rand(B, N, generator=g, **f32) * 50 - 25
    ctx[..., 1] = a + b * ctx[..., 0] + torch.randn(B, N, generator=g, **f32) * 0.5
    ctx[..., 2] = torch.randn(B, N, generator=g, **f32) * 0.1
    if C > 3:
        u = torch.rand(B, N, generator=g, **f32).clamp_min(1e-7)
        ctx[..., 3] = (-12.0 * torch.log(u)).round().clamp(0, 255)
    if C > 4:
        ctx[..., 4:] = torch.randn(B, N, C - 4, generator=g, **f32)
    yaw = (torch.rand(B, 1, generator=g, **f32) * 4 - 2) * (math.pi / 180)
    noisy = gt.clone()
    noisy[..., 0] = gt[..., 0] * torch.cos(yaw) - gt[..., 1] * torch.sin(yaw)
    noisy[..., 1] = gt[..., 0] * torch.sin(yaw) + gt[..., 1] * torch.cos(yaw)
    noisy[..., 0] += torch.rand(B, 1, generator=g, **f32) * 0.8 - 0.4
    noisy[..., 1] += torch.rand(B, 1, generator=g, **f32) * 0.8 - 0.4
    noisy[..., 2] += torch.rand(B, 1, generator=g, **f32) * 0.2 - 0.1
    noisy += torch.randn(B, M, 3, generator=g, **f32) * 0.05
    target = gt - noisy
    center = noisy.mean(dim=1, keepdim=True)                  # src/dataset.py:229-235
    ctx[..., :3] -= center
    noisy = noisy - center
    return ctx.contiguous(), noisy.contiguous(), target.contiguous()
