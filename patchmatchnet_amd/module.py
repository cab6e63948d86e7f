"""Layer primitives of the PatchmatchNet boundary (mirror of the reference's ``models/module.py`` interface).

``ConvBnReLU`` (2-D) runs on PyTorch-ROCm / MIOpen.  ``ConvBnReLU3D`` only ever appears as a 1x1x1 pointwise layer
inside the three tiny MLPs of the hot path; here it is a parameter container (same state-dict names) whose arithmetic
is executed inside the HIP kernels (see ``params.pack_mlp``).  ``differentiable_warping`` is the HIP op.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class ConvBnReLU(nn.Module):
    """Conv2d (no bias) + BatchNorm2d + ReLU; reference models/module.py:11-40."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, pad: int = 1,
                 dilation: int = 1) -> None:
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, dilation=dilation,
                              bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.relu(self.bn(self.conv(x)), inplace=True)


class ConvBnReLU3D(nn.Module):
    """Conv3d (no bias) + BatchNorm3d + ReLU; reference models/module.py:43-72.

    Parameter container: on the hot path these layers are 1x1x1 and are evaluated inside the fused kernels."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, pad: int = 1,
                 dilation: int = 1) -> None:
        super().__init__()
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, dilation=dilation,
                              bias=False)
        self.bn = nn.BatchNorm3d(out_channels)

    def bn_tensors(self):
        return (self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover - not used by the fused path
        return F.relu(self.bn(self.conv(x)), inplace=True)


def differentiable_warping(src_fea: torch.Tensor, src_proj: torch.Tensor, ref_proj: torch.Tensor,
                           depth_samples: torch.Tensor) -> torch.Tensor:
    """Homography warp + bilinear gather; same signature / result as reference models/module.py:130-181
    (inference only -- no autograd)."""
    return ops.differentiable_warping(src_fea, src_proj, ref_proj, depth_samples)


def depth_regression(p: torch.Tensor, depth_values: torch.Tensor) -> torch.Tensor:
    """Expectation of depth_values [B,D] under p [B,D,H,W] -> [B,1,H,W]; reference models/module.py:184-196."""
    return torch.sum(p * depth_values.view(depth_values.shape[0], -1, 1, 1), dim=1).unsqueeze(1)


def is_empty(x: torch.Tensor) -> bool:
    return x.numel() == 0
