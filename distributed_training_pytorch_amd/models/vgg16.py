"""VGG16 (configuration D) on the fused HIP ops, checkpoint-compatible with the reference.

Parity target: reference ``model/vgg16.py:5-80`` — five ``ConvBlock``s (2,2,3,3,3 conv3x3+ReLU,
each block ending in MaxPool 2x2), ``AdaptiveAvgPool2d((7,7))``, flatten, ``linear1`` 25088->4096,
``linear2`` 4096->4096, ``linear3`` 4096->out_channels, one shared ReLU and Dropout(0.3); Kaiming
(fan_out) conv init, N(0, 0.01) linear init, zero biases (``:49-57``).

``state_dict`` keys are identical (``backbone.{b}.conv.{i}.weight|bias``, ``linear{1,2,3}.*``;
SURVEY.md §2.3): convolutions sit at the even indices of ``ConvBlock.conv`` and a parameter-less
``ReLUMarker`` keeps the odd indices occupied, because on the GPU the ReLU is an epilogue of the
conv kernel.  Values are emitted as contiguous fp32 NCHW tensors, so a stock ``torch.nn`` VGG16
(or the reference's ``eval.py``) loads them unchanged.

Differences on purpose: ``in_channels`` is honoured (reference ignores it, quirk #22); GPU compute
is bf16 with f32 accumulation against fp32 master weights held in a flat ``ParamArena``.
"""
import os

import torch
import torch.nn as nn

from ..ops.layers import (AvgPoolFlatten, BatchNormReLU, Conv3x3, ConvFirst, Linear, MaxPool2x2, ReLUMarker,
                          bn_relu_pool, conv_bn_relu_folded, conv_relu_pool)
from .fused_model import FusedModel


SELF_MASK = os.environ.get("DTP_CONV1_SELF_MASK", "1") == "1"     # A/B switch (see ConvBlock)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers=2, first=False, batch_norm=False):
        super().__init__()
        layers = []
        for i in range(num_layers):
            cin = in_channels if i == 0 else out_channels
            fused_relu = not batch_norm
            if first and i == 0:
                # conv1_1 masks its own incoming gradient inside its weight-gradient kernel (it
                # recomputes the forward sign from the image), so conv1_2's data-gradient does not
                # read the 64-channel full-resolution activation as a mask
                conv = ConvFirst(cin, out_channels, relu=fused_relu, grad_premasked=fused_relu and not SELF_MASK,
                                 self_mask=fused_relu and SELF_MASK)
            else:
                # the input of a block's first conv is a pool output: its mask is applied by the
                # pool backward, so only convs fed directly by a fused conv+ReLU mask their dgrad
                conv = Conv3x3(cin, out_channels, relu=fused_relu,
                               input_is_relu_out=(i > 0 and fused_relu and not (first and SELF_MASK)),
                               grad_premasked=fused_relu,
                               # 1-bit (y > 0) masks for the next conv's data gradient; a block's last
                               # conv feeds the pool, whose codes already carry the sign
                               emit_relu_bits=fused_relu and i < num_layers - 1)
            conv.bn_follows = bool(batch_norm)       # its bias gradient is identically zero in training
            layers.append(conv)
            if batch_norm:
                # conv -> BatchNorm -> ReLU: normalisation and ReLU are one hand-written kernel pair
                # (csrc/bn.hip); the marker keeps the Sequential indices of a torch VGG16-BN
                layers.append(BatchNormReLU(out_channels))
            layers.append(ReLUMarker())
        layers.append(MaxPool2x2(relu_mask=not batch_norm))
        self.conv = nn.Sequential(*layers)
        self.batch_norm = batch_norm

    def forward(self, x):
        if x.is_cuda and not self.batch_norm:
            # last conv + ReLU + pool as one kernel where the shapes allow it (block 1: the
            # 1.6 GB pre-pool activation at batch 256 is neither written nor re-read)
            mods = list(self.conv)
            for m in mods[:-3]:
                x = m(x)
            y = conv_relu_pool(mods[-3], x) if isinstance(mods[-1], MaxPool2x2) and mods[-1].relu_mask else None
            if y is not None:
                return y
            for m in mods[-3:]:
                x = m(x)
            return x
        if x.is_cuda and self.batch_norm and not self.training and not torch.is_grad_enabled():
            # inference: each conv + BatchNorm + ReLU triple is ONE convolution with folded weights
            mods = list(self.conv)
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, (Conv3x3, ConvFirst)) and i + 1 < len(mods) and isinstance(mods[i + 1], BatchNormReLU):
                    # the block's last triple is followed by (ReLU marker, MaxPool): the pool joins the kernel
                    last = (i + 4 == len(mods) and isinstance(mods[i + 2], ReLUMarker) and not mods[i + 2].active
                            and isinstance(mods[i + 3], MaxPool2x2))
                    y = conv_bn_relu_folded(m, mods[i + 1], x, pool=last)
                    if y is not None:
                        if last:
                            x, pooled = y
                            if pooled:
                                return x
                        else:
                            x = y
                        i += 2
                        continue
                x = m(x)
                i += 1
            return x
        if x.is_cuda and self.batch_norm and self.training:
            # training: the block's last BatchNorm + ReLU + max-pool is one kernel pair — the
            # normalised full-resolution activation and its gradient never reach HBM
            mods = list(self.conv)
            if isinstance(mods[-1], MaxPool2x2) and isinstance(mods[-3], BatchNormReLU):
                for m in mods[:-3]:
                    x = m(x)
                y = bn_relu_pool(mods[-3], x)
                if y is not None:
                    return y
                for m in mods[-3:]:
                    x = m(x)
                return x
        return self.conv(x)


class VGG16(FusedModel):
    def __init__(self, in_channels=3, out_channels=1, init_weights=True, batch_norm=False,
                 dropout=0.3, grad_dtype=torch.float32, fp8_classifier=False):
        super().__init__(grad_dtype=grad_dtype)
        self.fp8_classifier = bool(fp8_classifier)   # linear1/linear2 forward GEMMs in OCP e4m3
        self.batch_norm = batch_norm
        self.backbone = nn.Sequential(ConvBlock(in_channels, 64, first=True, batch_norm=batch_norm),
                                      ConvBlock(64, 128, batch_norm=batch_norm),
                                      ConvBlock(128, 256, num_layers=3, batch_norm=batch_norm),
                                      ConvBlock(256, 512, num_layers=3, batch_norm=batch_norm),
                                      ConvBlock(512, 512, num_layers=3, batch_norm=batch_norm))
        self.avgpool = AvgPoolFlatten((7, 7))
        self.linear1 = Linear(512 * 7 * 7, 4096, relu=True, drop_p=dropout, grad_premasked=True,
                              fp8=fp8_classifier)
        self.linear2 = Linear(4096, 4096, relu=True, drop_p=dropout, input_is_relu_out=True,
                              input_drop_p=dropout, grad_premasked=True, fp8=fp8_classifier)
        # kept for attribute parity with the reference (model/vgg16.py:39-40); fused on the GPU
        self.relu = ReLUMarker()
        self.dropout = nn.Dropout(dropout)
        self.linear3 = Linear(4096, out_channels, out_fp32=True, input_is_relu_out=True,
                              input_drop_p=dropout)
        if init_weights:
            self._initialize_weights()
        self.finalize()

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, (Conv3x3, ConvFirst)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    # ------------------------------------------------------------------ input normalisation
    def _after_rehome(self):
        if getattr(self, "input_norm", None) is not None:
            self.set_input_normalization(*self.input_norm)      # constants follow the device

    def set_input_normalization(self, scale=1.0, mean=None, std=None):
        """Accept raw (e.g. uint8) images: ``(x*scale - mean)/std`` is applied inside the first
        convolution's ingest kernel on the GPU (a plain tensor op on the CPU path)."""
        self.input_norm = (float(scale), None if mean is None else tuple(mean), None if std is None else tuple(std))
        self.backbone[0].conv[0].set_input_normalization(scale, mean, std)

    def forward(self, x):
        x = self.prepare_input(x)
        x = self.backbone(x)
        x = self.avgpool(x)
        x = self.linear1(x)      # + ReLU + dropout fused
        x = self.linear2(x)
        return self.finish_forward(self.linear3(x))


def vgg16_bn(in_channels=3, out_channels=1, **kw):
    return VGG16(in_channels, out_channels, batch_norm=True, **kw)
