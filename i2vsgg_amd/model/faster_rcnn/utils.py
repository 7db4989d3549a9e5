"""``FC`` / ``Conv2d`` helper blocks of the relation head (faster_rcnn/utils.py:32-58): Linear / Conv
followed by ReLU, with the reference's sub-module names (``.fc``, ``.conv``) so state_dict keys match
(``vrd.fc6.fc.weight``, ``vrd.conv_lo.0.conv.bias``).  Both run on the implicit-GEMM kernel with the
bias and ReLU fused into the epilogue."""
import math

import torch
import torch.nn as nn

from i2vsgg_amd import ops
from .layers import ConvParams


class _LinearParams(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        bound = 1.0 / math.sqrt(in_features)
        self.weight = nn.Parameter(torch.empty(out_features, in_features).uniform_(-bound, bound))
        self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))


class Linear(_LinearParams):
    """nn.Linear replacement (same ``weight`` / ``bias`` keys) running on the implicit-GEMM kernel."""

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class FC(nn.Module):
    def __init__(self, in_features, out_features, relu=True):
        super().__init__()
        self.fc = _LinearParams(in_features, out_features)
        self.relu = relu

    def forward(self, x):
        return ops.linear(x, self.fc.weight, self.fc.bias, relu=self.relu)


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, same_padding=False, bn=False):
        super().__init__()
        assert not bn, "the reference instantiates these blocks with bn=False (resnet_SGG_emb.py:65,104-106)"
        pad = int((kernel_size - 1) / 2) if same_padding else 0
        k = kernel_size
        self.conv = ConvParams(in_channels, out_channels, k, stride, pad, bias=True,
                               std=1.0 / math.sqrt(3.0 * in_channels * k * k))
        self.relu = relu

    def forward(self, x):
        c = self.conv
        w = c.weight
        if c.cin % 4:       # e.g. the 2-channel dual mask: pad the channel axis to a float4 boundary
            padc = 4 - c.cin % 4
            if x.shape[1] == c.cin:                      # a caller may hand the input already padded with zero channels
                x = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, padc))
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, padc))
        return ops.conv2d(x, w, None, c.bias, None, c.stride, c.pad, relu=self.relu)
