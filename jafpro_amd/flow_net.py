"""Appearance-propagation network on the HIP kernels.  Mirrors src/flow_net.py:
CompositeWeightUnet (:6-58), Propagation3DFlowNet (:61-99), ResnetBlock (:102-141) with the
reference's Sequential indices, so the state_dict keys match.

Only the configuration stage 4 constructs is implemented in HIP:
``Propagation3DFlowNet(9, 32, 2, 3, use_deconv=False)`` -- batch norm, ReLU, reflection padding,
bilinear x2 (align_corners=False) upsampling.  Everything else raises NotImplementedError.
BatchNorm follows ``self.training`` exactly as nn.BatchNorm2d does; note that the reference's
inference script never calls .eval() on this module.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .networks import _BN
from .ops import ACT_NONE, ACT_RELU, ACT_SIGMOID


class _Conv2d(nn.Module):
    def __init__(self, cin, cout, k, stride=1, pad=0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / math.sqrt(cin * k * k)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)
        self.stride, self.pad = stride, pad

    def forward(self, x, act=ACT_NONE):
        return ops.conv2d(x, self.weight, self.bias, stride=self.stride, pad=self.pad, act=act)


class _Slot(nn.Module):
    """Parameter-free placeholder keeping the reference's Sequential numbering."""

    def __init__(self, kind, arg=None):
        super().__init__()
        self.kind, self.arg = kind, arg


def _run(seq: nn.Sequential, x, training: bool):
    """Interprets a reference-shaped Sequential: pad / conv / bn(+act) / up / resblock / sigmoid."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, _Slot) and m.kind == "pad":
            x = ops.reflect_pad(x, m.arg)
        elif isinstance(m, _Slot) and m.kind == "up":
            x = ops.resize(x, (x.shape[2] * 2, x.shape[3] * 2), align_corners=False)
        elif isinstance(m, _Conv2d):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, _Slot) and nxt.kind == "sigmoid":
                x = m(x, ACT_SIGMOID)
                i += 1
            else:
                x = m(x)
        elif isinstance(m, _BN):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(nxt, _Slot) and nxt.kind == "act":
                x = m.apply_bn(x, ACT_RELU, 0.0, training)
                i += 1
            else:
                x = m.apply_bn(x, ACT_NONE, 0.0, training)
        elif isinstance(m, ResnetBlock):
            x = m(x)
        elif isinstance(m, _Slot) and m.kind == "act":
            raise RuntimeError("activation slot without a producer")
        i += 1
    return x


class ResnetBlock(nn.Module):
    def __init__(self, dim, padding_type, norm_layer=None, activation=None, use_dropout=False):
        super().__init__()
        if padding_type != 'reflect' or use_dropout:
            raise NotImplementedError("stage 4 uses reflect padding without dropout")
        self.conv_block = nn.Sequential(
            _Slot("pad", 1), _Conv2d(dim, dim, 3), _BN(dim), _Slot("act"),
            _Slot("pad", 1), _Conv2d(dim, dim, 3), _BN(dim))

    def forward(self, x):
        cb = self.conv_block
        y = cb[1](ops.reflect_pad(x, 1))
        y = cb[2].apply_bn(y, ACT_RELU, 0.0, self.training)
        y = cb[5](ops.reflect_pad(y, 1))
        return cb[6].apply_bn(y, ACT_NONE, 0.0, self.training, residual=x)     # x + conv_block(x)


class CompositeWeightUnet(nn.Module):
    def __init__(self, input_nc, ngf, n_downsampling, n_blocks, norm_layer=None, act=None,
                 padding_type='reflect', use_deconv=False, use_tgt_dp=False):
        super().__init__()
        if use_deconv:
            raise NotImplementedError("ConvTranspose upsampling is not on the stage-4 path (use_deconv=False)")
        input_nc = input_nc + 3 * use_tgt_dp
        down = [_Slot("pad", 3), _Conv2d(input_nc, ngf, 7), _BN(ngf), _Slot("act")]
        for i in range(n_downsampling):
            mult = 2 ** i
            down += [_Conv2d(ngf * mult, ngf * mult * 2, 3, stride=2, pad=1), _BN(ngf * mult * 2), _Slot("act")]
        mult = 2 ** n_downsampling
        for _ in range(n_blocks - n_blocks // 2):
            down += [ResnetBlock(ngf * mult, padding_type)]
        res = [ResnetBlock(ngf * mult, padding_type) for _ in range(n_blocks // 2)]
        up = []
        for i in range(n_downsampling):
            mult = 2 ** (n_downsampling - i)
            up += [_Slot("up"), _Conv2d(ngf * mult, ngf * mult // 2, 3, pad=1), _BN(ngf * mult // 2), _Slot("act")]
        final = [_Slot("pad", 3), _Conv2d(ngf, 1, 7), _Slot("sigmoid")]
        self.model_down_img = nn.Sequential(*down)
        self.model_res_img = nn.Sequential(*res)
        self.model_up_img = nn.Sequential(*up)
        self.model_final_w = nn.Sequential(*final)

    def forward(self, input):
        x = _run(self.model_down_img, input, self.training)
        x = _run(self.model_res_img, x, self.training)
        x = _run(self.model_up_img, x, self.training)
        return _run(self.model_final_w, x, self.training)


class Propagation3DFlowNet(nn.Module):
    def __init__(self, input_nc, ngf, n_downsampling, n_blocks, norm_type='batch', act_type='relu',
                 padding_type='reflect', use_deconv=True, use_tgt_dp=False):
        super().__init__()
        if norm_type != 'batch' or act_type != 'relu':
            raise NotImplementedError("stage 4 uses norm_type='batch', act_type='relu'")
        self.composite_unet = CompositeWeightUnet(input_nc, ngf, n_downsampling, n_blocks, None, None,
                                                  padding_type, use_deconv, use_tgt_dp)
        self.use_tgt_dp = use_tgt_dp

    def forward(self, x):
        fake_tgt, tsf_image, tgt_IUV, use_IUV = x['fake_tgt'], x['tsf_image'], x['tgt_IUV'], x['use_IUV']
        use_mask, tgt_smpl_mask = x['use_mask'], x['tgt_smpl_mask']
        fake_tgt = fake_tgt.contiguous()
        tsf_image = tsf_image.contiguous()
        if use_mask:
            tsf_image = ops.mul_bcast(tsf_image, tgt_smpl_mask.contiguous())     # :91
        srcs = [tsf_image, fake_tgt] + ([tgt_IUV.contiguous()] if use_IUV else [])
        # the first layer is ReflectionPad(3)+conv7: pad each source, the 9-channel cat is never built
        cu = self.composite_unet
        down = list(cu.model_down_img)
        padded = [ops.reflect_pad(s, 3) for s in srcs]
        h = down[1](padded)
        h = down[2].apply_bn(h, ACT_RELU, 0.0, cu.training)
        h = _run(nn.Sequential(*down[4:]), h, cu.training)
        h = _run(cu.model_res_img, h, cu.training)
        h = _run(cu.model_up_img, h, cu.training)
        weight = _run(cu.model_final_w, h, cu.training)
        pred = ops.blend(fake_tgt, tsf_image, weight)                            # :98
        return {'pred_target': pred, 'weight': weight}
