"""FlowNetSD on the HIP convolution kernels: the temporal-consistency term of the evaluation script
(test/video_evaluation.py:66-67,197-206: `flow_criterion = FlowNetSD(args=[], batchNorm=False)`, fed
`flownet_preprocess(cat[prev, cur])`, output `[0]` = flow2, L1 between the flows of the predicted and the real video).

Mirrors src/flownet2_pytorch/networks/FlowNetSD.py:11-106 and submodules.py:7-38 -- same attribute names and
`state_dict` keys / shapes (conv0.0.weight ... upsampled_flow3_to_2.bias), so the FlowNet2-SD checkpoint's `state_dict` loads
unchanged -- with conv + bias + LeakyReLU(0.1) and ConvTranspose2d(4, 2, 1) + bias [+ LeakyReLU(0.1)] as single launches and
the encoder/decoder concatenations read in place by the three-source convolution.  Inference only; the
batch-norm variant (batchNorm=True) is not what the evaluation constructs and is not built.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


class _Conv(nn.Module):
    """nn.Conv2d(cin, cout, k, stride, (k-1)//2, bias=True) parameters; applied by the parent with a fused activation."""

    def __init__(self, cin, cout, k=3, stride=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        self.stride, self.pad = stride, (k - 1) // 2
        nn.init.xavier_uniform_(self.weight)             # FlowNetSD.py:51-55
        nn.init.uniform_(self.bias)

    def forward(self, srcs, act=ops.ACT_NONE, slope=0.0):
        return ops.conv2d(srcs, self.weight, self.bias, stride=self.stride, pad=self.pad, act=act, slope=slope)


class _Deconv(nn.Module):
    """nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True) parameters (weight [cin, cout, 4, 4])."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, cout, 4, 4))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.xavier_uniform_(self.weight)             # FlowNetSD.py:57-60
        nn.init.uniform_(self.bias)

    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        return ops.conv_transpose2d(x, self.weight, self.bias, 2, 1, act, slope)


def conv(batchNorm, in_planes, out_planes, kernel_size=3, stride=1):
    """submodules.py:7-19 without batch norm: Sequential(Conv2d(bias), LeakyReLU(0.1)) -- keys `<name>.0.weight/bias`."""
    if batchNorm:
        raise NotImplementedError("FlowNetSD(batchNorm=True) is not built (the evaluation uses batchNorm=False)")
    return nn.Sequential(_Conv(in_planes, out_planes, kernel_size, stride), nn.LeakyReLU(0.1, inplace=True))


def i_conv(batchNorm, in_planes, out_planes, kernel_size=3, stride=1, bias=True):
    if batchNorm:
        raise NotImplementedError("FlowNetSD(batchNorm=True) is not built")
    return nn.Sequential(_Conv(in_planes, out_planes, kernel_size, stride))


def predict_flow(in_planes):
    return _Conv(in_planes, 2, 3, 1)


def deconv(in_planes, out_planes):
    return nn.Sequential(_Deconv(in_planes, out_planes), nn.LeakyReLU(0.1, inplace=True))


class FlowNetSD(nn.Module):
    def __init__(self, args=None, batchNorm=False):
        super().__init__()
        self.batchNorm = batchNorm
        self.conv0 = conv(batchNorm, 6, 64)
        self.conv1 = conv(batchNorm, 64, 64, stride=2)
        self.conv1_1 = conv(batchNorm, 64, 128)
        self.conv2 = conv(batchNorm, 128, 128, stride=2)
        self.conv2_1 = conv(batchNorm, 128, 128)
        self.conv3 = conv(batchNorm, 128, 256, stride=2)
        self.conv3_1 = conv(batchNorm, 256, 256)
        self.conv4 = conv(batchNorm, 256, 512, stride=2)
        self.conv4_1 = conv(batchNorm, 512, 512)
        self.conv5 = conv(batchNorm, 512, 512, stride=2)
        self.conv5_1 = conv(batchNorm, 512, 512)
        self.conv6 = conv(batchNorm, 512, 1024, stride=2)
        self.conv6_1 = conv(batchNorm, 1024, 1024)

        self.deconv5 = deconv(1024, 512)
        self.deconv4 = deconv(1026, 256)
        self.deconv3 = deconv(770, 128)
        self.deconv2 = deconv(386, 64)

        self.inter_conv5 = i_conv(batchNorm, 1026, 512)
        self.inter_conv4 = i_conv(batchNorm, 770, 256)
        self.inter_conv3 = i_conv(batchNorm, 386, 128)
        self.inter_conv2 = i_conv(batchNorm, 194, 64)

        self.predict_flow6 = predict_flow(1024)
        self.predict_flow5 = predict_flow(512)
        self.predict_flow4 = predict_flow(256)
        self.predict_flow3 = predict_flow(128)
        self.predict_flow2 = predict_flow(64)

        self.upsampled_flow6_to_5 = _Deconv(2, 2)
        self.upsampled_flow5_to_4 = _Deconv(2, 2)
        self.upsampled_flow4_to_3 = _Deconv(2, 2)
        self.upsampled_flow3_to_2 = _Deconv(2, 2)
        self.upsample1 = nn.Upsample(scale_factor=4, mode='bilinear')      # built by the reference, never called in forward

    @staticmethod
    def _lrelu_conv(seq, srcs):
        return seq[0](srcs, ops.ACT_LRELU, 0.1)

    @staticmethod
    def _lrelu_deconv(seq, srcs):
        x = srcs[0] if len(srcs) == 1 else torch.cat(srcs, 1)              # the transposed convolution reads one tensor
        return seq[0](x.contiguous(), ops.ACT_LRELU, 0.1)

    @torch.no_grad()
    def forward(self, x):
        """x [N,6,H,W] (two RGB frames in [0,1], H and W multiples of 64) -> (flow2,) in eval mode, (flow2..flow6) in train
        mode -- the evaluation script leaves the module in train mode and takes `[0]`."""
        c, d = self._lrelu_conv, self._lrelu_deconv
        x = x.contiguous()
        out_conv0 = c(self.conv0, [x])
        out_conv1 = c(self.conv1_1, [c(self.conv1, [out_conv0])])
        out_conv2 = c(self.conv2_1, [c(self.conv2, [out_conv1])])
        out_conv3 = c(self.conv3_1, [c(self.conv3, [out_conv2])])
        out_conv4 = c(self.conv4_1, [c(self.conv4, [out_conv3])])
        out_conv5 = c(self.conv5_1, [c(self.conv5, [out_conv4])])
        out_conv6 = c(self.conv6_1, [c(self.conv6, [out_conv5])])

        flow6 = self.predict_flow6([out_conv6])
        flow6_up = self.upsampled_flow6_to_5(flow6)
        out_deconv5 = d(self.deconv5, [out_conv6])
        concat5 = [out_conv5, out_deconv5, flow6_up]
        flow5 = self.predict_flow5([self.inter_conv5[0](concat5)])

        flow5_up = self.upsampled_flow5_to_4(flow5)
        out_deconv4 = d(self.deconv4, concat5)
        concat4 = [out_conv4, out_deconv4, flow5_up]
        flow4 = self.predict_flow4([self.inter_conv4[0](concat4)])

        flow4_up = self.upsampled_flow4_to_3(flow4)
        out_deconv3 = d(self.deconv3, concat4)
        concat3 = [out_conv3, out_deconv3, flow4_up]
        flow3 = self.predict_flow3([self.inter_conv3[0](concat3)])

        flow3_up = self.upsampled_flow3_to_2(flow3)
        out_deconv2 = d(self.deconv2, concat3)
        concat2 = [out_conv2, out_deconv2, flow3_up]
        flow2 = self.predict_flow2([self.inter_conv2[0](concat2)])
        if self.training:
            return flow2, flow3, flow4, flow5, flow6
        return flow2,


def flownet_preprocess(img_pair: torch.Tensor) -> torch.Tensor:
    """test/video_evaluation.py:34-36: (-1, 1) -> (0, 1)."""
    return ops.axpby(0.5, img_pair.contiguous(), 0.5, torch.ones_like(img_pair))
