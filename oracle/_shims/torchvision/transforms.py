"""See package docstring.  Lives in a real file because the reference wraps these
modules in `torch.jit.script` (utils/transforms.py:32-37), which needs source."""
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class Resize(nn.Module):
    def __init__(self, size: List[int]):
        super().__init__()
        self.size = size

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[-2] == self.size[0] and x.shape[-1] == self.size[1]:
            return x
        # torchvision's tensor Resize default: bilinear + antialias
        squeeze = x.dim() == 3
        if squeeze:
            x = x.unsqueeze(0)
        x = F.interpolate(x, size=self.size, mode="bilinear", align_corners=False, antialias=True)
        return x.squeeze(0) if squeeze else x


class Normalize(nn.Module):
    def __init__(self, mean: List[float], std: List[float]):
        super().__init__()
        self.mean = mean
        self.std = std

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        mean = torch.tensor(self.mean, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        std = torch.tensor(self.std, dtype=x.dtype, device=x.device).view(-1, 1, 1)
        return (x - mean) / std


class ToTensor:
    def __call__(self, pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)
        if t.dtype == torch.uint8:
            t = t.float().div(255)
        return t
