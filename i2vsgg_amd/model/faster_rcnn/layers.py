"""Parameter holders and the ResNet C4 building blocks, executed on the HIP implicit-GEMM kernels.

Module / parameter NAMES reproduce the reference's state_dict keys (``RCNN_base.6.3.conv2.weight``,
``RCNN_base.1.running_var`` ...), so reference checkpoints load with ``load_state_dict``.  Conv
weights keep the logical (Cout,Cin,KH,KW) shape but live in channels_last memory, which IS the
K-major (Cout,KH,KW,Cin) filter layout of the GEMM.  Frozen BatchNorm (eval mode everywhere in the
reference: resnet_instance_styleD_bilinear.py:405-411,433-439) is a per-channel scale/shift fused
into the conv epilogue together with the residual add and the ReLU."""
import math
import os

import torch
import torch.nn as nn

from i2vsgg_amd import ops

_CL = torch.channels_last


class ConvParams(nn.Module):
    """Holds ``weight`` (and ``bias``) like nn.Conv2d; compute happens in the caller's fused op."""

    def __init__(self, cin, cout, k, stride=1, pad=0, bias=False, std=None):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        w = torch.empty(cout, cin, k, k)
        w.normal_(0, std if std is not None else math.sqrt(2.0 / (k * k * cout)))   # ResNet.__init__ :238-241
        self.weight = nn.Parameter(w.contiguous(memory_format=_CL))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        with torch.no_grad():       # keep the K-major filter layout across .cuda()/.to()
            if not self.weight.is_contiguous(memory_format=_CL):
                self.weight.data = self.weight.data.contiguous(memory_format=_CL)
        return self

    def forward(self, x, relu=False):
        return ops.conv2d(x, self.weight, None, self.bias, None, self.stride, self.pad, relu=relu)


class FrozenBN(nn.Module):
    """Eval-mode BatchNorm2d as scale/shift vectors; parameters never require grad."""

    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(c), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(c), requires_grad=False)
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._folded = None

    def folded(self):
        f = self._folded
        if f is None or f[0].device != self.weight.device:
            with torch.no_grad():
                scale = self.weight / torch.sqrt(self.running_var + self.eps)
                shift = self.bias - self.running_mean * scale
            self._folded = f = (scale.contiguous(), shift.contiguous())
        return f

    def invalidate(self):
        self._folded = None

    def _load_from_state_dict(self, *a, **k):
        self._folded = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._folded = None
        return super()._apply(fn, *a, **k)


# I2V_WINOGRAD: 0 = direct 3x3, 2 = F(2x2,3x3) (fp32 error ~1e-6 like the direct kernel), 4 (default) = F(4x4,3x3) (4x fewer
# MACs, error ~1e-5 per layer: what cuDNN's fp32 WINOGRAD_NONFUSED does for the reference's 3x3 layers)
WINOGRAD = int(os.environ.get("I2V_WINOGRAD", "4"))
WINOGRAD_MIN_CIN = {0: 1 << 30, 2: 128, 4: 64}[WINOGRAD]      # measured: F(2x2) loses to direct at 64 channels, F(4x4) wins


class Bottleneck(nn.Module):
    """resnet_instance_styleD_bilinear.py:181-217: stride on the first 1x1 (caffe style)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ConvParams(inplanes, planes, 1, stride)
        self.bn1 = FrozenBN(planes)
        self.conv2 = ConvParams(planes, planes, 3, 1, 1)
        self.bn2 = FrozenBN(planes)
        self.conv3 = ConvParams(planes, planes * 4, 1)
        self.bn3 = FrozenBN(planes * 4)
        self.downsample = downsample
        self.stride = stride
        # the backward protocol of ops._BottleneckFn: my input is the previous block's ReLU output / the next block (the
        # only consumer of my output) masks the gradient it hands me.  Set by make_layer.
        self.in_relu = False
        self._next = []

    def _fused(self):
        """The whole block as one autograd node: gradients being recorded, every filter training, no per-filter fused SGD
        registered for them (strided blocks included: their stride sits on the two 1x1 layers reading the block input)."""
        if not (ops.BLOCK_FUSED and torch.is_grad_enabled()) or (self.stride != 1 and self.downsample is None):
            return False
        ws = [self.conv1.weight, self.conv2.weight, self.conv3.weight] + ([self.downsample[0].weight] if self.downsample is not None else [])
        return all(w.requires_grad and w.data_ptr() not in ops.FUSED_SGD for w in ws) and self.conv1.cin % 4 == 0

    def _winograd_filter(self):
        w = self.conv2.weight
        key = ops.param_key(w)
        if getattr(self, "_wino_key", None) != key:
            with torch.no_grad():
                self._wino_u, self._wino_key = ops.winograd_filter(w.detach(), WINOGRAD), key
        return self._wino_u

    def forward(self, x, out=None):
        """``out`` (forward-only callers): the block's result is written into this channels_last tensor."""
        s1, b1 = self.bn1.folded()
        s2, b2 = self.bn2.folded()
        s3, b3 = self.bn3.folded()
        if out is None and self._fused() and x.is_cuda:
            down = None
            if self.downsample is not None:
                down = (self.downsample[0].weight,) + tuple(self.downsample[1].folded())
            nxt = self._next[0] if self._next else None
            return ops.bottleneck(x, self.conv1.weight, self.conv2.weight, self.conv3.weight, (s1, b1), (s2, b2), (s3, b3), down,
                                  in_relu=self.in_relu, out_premasked=nxt is not None and nxt._fused(), stride=self.stride)
        h = ops.conv2d(x, self.conv1.weight, s1, b1, None, self.stride, 0, relu=True)
        if WINOGRAD and not torch.is_grad_enabled() and self.conv2.cin >= WINOGRAD_MIN_CIN:
            # no gradient is being recorded (the detached SGG_emb backbone, eval): the 3x3 runs as Winograd F(2x2,3x3)
            # with the filter transformed once -- 4x fewer MACs with F(4x4,3x3) (layer3: 63 -> 37 us per layer)
            h = ops.conv3x3_winograd(h, self._winograd_filter(), s2, b2, relu=True)
        else:
            h = ops.conv2d(h, self.conv2.weight, s2, b2, None, 1, 1, relu=True, winograd=bool(WINOGRAD))   # trained: F(4x4) fwd + dgrad
        res = x
        if self.downsample is not None:
            sd, bd = self.downsample[1].folded()
            res = ops.conv2d(x, self.downsample[0].weight, sd, bd, None, self.stride, 0, relu=False)
        return ops.conv2d(h, self.conv3.weight, s3, b3, res, 1, 0, relu=True, out=out)   # +residual, ReLU fused


def make_layer(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes * 4:
        down = nn.Sequential(ConvParams(inplanes, planes * 4, 1, stride), FrozenBN(planes * 4))
    layers = [Bottleneck(inplanes, planes, stride, down)]
    layers += [Bottleneck(planes * 4, planes) for _ in range(1, blocks)]
    for prev, blk in zip(layers[:-1], layers[1:]):      # inside a layer a block's output has exactly one consumer: the next block
        blk.in_relu = True
        prev._next = [blk]                              # (a list: not a registered submodule)
    return nn.Sequential(*layers), planes * 4


class _Marker(nn.Module):
    """Parameter-free slot so that Sequential indices match the reference (relu = 2, maxpool = 3)."""

    def forward(self, x):
        return x


class C4Base(nn.Sequential):
    """``RCNN_base`` = conv1, bn1, relu, maxpool, layer1, layer2, layer3 (same child indices as
    resnet_instance_styleD_bilinear.py:372-373).  The stem conv+BN+ReLU is one fused kernel on a
    4-channel padded input (Cin % 4 == 0 keeps every im2col load a 16-byte access)."""

    def __init__(self, blocks=(3, 4, 23)):
        conv1 = ConvParams(3, 64, 7, 2, 3)
        bn1 = FrozenBN(64)
        l1, c = make_layer(64, 64, blocks[0], 1)
        l2, c = make_layer(c, 128, blocks[1], 2)
        l3, c = make_layer(c, 256, blocks[2], 2)
        # layer1's output has one consumer, layer2's first block: the pre-masked hand-over of the block backward crosses that
        # boundary too.  layer2's output does not (netD_style taps it, forward(tap=True)), nor does layer3's (RPN, ROI pooling)
        l2[0].in_relu = True
        l1[-1]._next = [l2[0]]
        super().__init__(conv1, bn1, _Marker(), _Marker(), l1, l2, l3)
        self.out_channels = c
        self._w4 = None

    def _stem_weight(self):
        w = self[0].weight
        key = ops.param_key(w)
        if self._w4 is None or self._w4_key != key:
            with torch.no_grad():
                w4 = torch.zeros((w.shape[0], 4, w.shape[2], w.shape[3]), device=w.device)
                w4[:, :3] = w
            self._w4, self._w4_key = w4.contiguous(memory_format=_CL), key
        return self._w4

    def stem(self, im):
        B, C, H, W = im.shape
        if C == 4:                    # the device front-end (ops.image_prep) already emits the zero-padded NHWC blob
            x4 = ops.as_nhwc(im)
        else:
            x4 = torch.zeros((B, 4, H, W), device=im.device, dtype=torch.float32).contiguous(memory_format=_CL)
            x4[:, :3] = im
        s, b = self[1].folded()
        x = ops.conv2d(x4, self._stem_weight(), s, b, None, 2, 3, relu=True)
        return ops.maxpool3x3s2(x)

    def forward(self, im, tap=False, out=None):
        """``out`` (no gradient being recorded): the C4 map is written into this channels_last tensor by the last layer."""
        x = self.stem(im)
        x = self[4](x)
        feat1 = self[5](x)
        if out is None:
            feat = self[6](feat1)
        else:
            blocks = list(self[6])
            feat = feat1
            for blk in blocks[:-1]:
                feat = blk(feat)
            feat = blocks[-1](feat, out=out)
        return (feat, feat1) if tap else feat

    # The same pass in two halves (train.SGGEmbStep's stage-split schedule, round 6): stem .. layer3[:cut] -> ``out`` (any
    # channels_last tensor of that block's output shape), and layer3[cut:] from there -> the C4 map.  0 < cut < len(layer3).
    def forward_front(self, im, cut, out=None):
        x = self[5](self[4](self.stem(im)))
        blocks = list(self[6])[:cut]
        for blk in blocks[:-1]:
            x = blk(x)
        return blocks[-1](x, out=out) if out is not None else blocks[-1](x)

    def forward_back(self, mid, cut, out=None, end=None):
        blocks = list(self[6])[cut:end]
        x = mid
        for blk in blocks[:-1]:
            x = blk(x)
        return blocks[-1](x, out=out) if out is not None else blocks[-1](x)


def load_reference_state(module, state_dict, strict=False):
    """load_state_dict that tolerates NCHW-contiguous checkpoint tensors (copy_ keeps OUR layout)."""
    out = module.load_state_dict(state_dict, strict=strict)
    for m in module.modules():
        if isinstance(m, FrozenBN):
            m.invalidate()
    return out
