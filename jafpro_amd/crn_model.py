"""CRN refinement / background network on the HIP kernels.  Mirrors src/crn_model.py:
LayerNorm (:67-87), ConvBlock (:90-106), CRN_smaller (:243-308) -- same constructor arguments,
``forward(label, sp)`` contract and state_dict keys (``conv1_encoder.conv_block.{0,3}.weight|bias``,
``.{1,4}.gamma|beta``, ``out_conv``, ``fg_conv``).

The decoder inputs ``cat[label_down, pool_k, net_up]`` (:276-299) are never materialised: the
conv kernel reads its three sources directly.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE


class _Conv2d(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout))
        bound = 1.0 / math.sqrt(cin * k * k)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class LayerNorm(nn.Module):
    """Per-sample mean / Bessel-corrected std over C*H*W, eps added to the std, per-channel
    gamma (init U(0,1)) and beta; fused with the following LeakyReLU(0.01) by ConvBlock."""

    def __init__(self, num_features, eps=1e-5, affine=True):
        super().__init__()
        self.num_features = num_features
        self.affine = affine
        self.eps = eps
        if not affine:
            raise NotImplementedError("the stage-4 CRN always uses the affine LayerNorm")
        self.gamma = nn.Parameter(torch.Tensor(num_features).uniform_())
        self.beta = nn.Parameter(torch.zeros(num_features))
        self.pre_stats = ops.LNStats()

    def forward(self, x, slope: float = 1.0, pre=None, dst=None, keep_f32=True, sole=False):
        """slope=1.0 is the bare LayerNorm; ConvBlock passes the LeakyReLU slope 0.01, the statistics
        its convolution's epilogue already accumulated (ops.LNStats) and, on the packed bf16 path, the packed image of
        the convolution that consumes the result (ops.PackedDst); sole: x is a convolution's output that nothing else reads
        (its gradient may then go back to that convolution as a packed bf16 image)."""
        return ops.layernorm_lrelu(x if x.is_contiguous() else x.contiguous(), self.gamma, self.beta, self.eps, slope, pre, dst, keep_f32, sole)


class ConvBlock(nn.Module):
    def __init__(self, n_repeats, c_in, c_out, kernel_size, pad):
        super().__init__()
        k = kernel_size[0] if isinstance(kernel_size, (tuple, list)) else kernel_size
        layers = []
        for _ in range(n_repeats):
            layers += [_Conv2d(c_in, c_out, k), LayerNorm(c_out), nn.Identity()]   # Identity = LeakyReLU slot
            c_in = c_out
        self.conv_block = nn.Sequential(*layers)
        self.n_repeats, self.pad = n_repeats, pad

    def forward(self, x, out_image=None):
        """x: tensor or list of tensors (read as their channel concatenation).  On the packed bf16 path each LayerNorm
        writes its result straight into the packed input image of the convolution that follows it (the next repeat's, or
        `out_image`: a single-source consumer of the block's output such as the CRN's 1x1 heads)."""
        img_in = None
        for r in range(self.n_repeats):
            conv, ln = self.conv_block[3 * r], self.conv_block[3 * r + 1]
            st = ln.pre_stats          # this LayerNorm's own side channel (zeroed once, kept clean by the finalize kernel)
            # bf16 storage (bf16 arithmetic mode): the pre-LayerNorm output is read by this block's LayerNorm alone, which takes
            # its statistics from the convolution's epilogue (unrounded) and reads the tensor in bf16
            x = ops.conv2d(x, conv.weight, conv.bias, stride=1, pad=self.pad, act=ACT_NONE, ln_stats=st, prepacked=img_in,
                           out_dtype=torch.bfloat16)
            img_out = out_image
            if r + 1 < self.n_repeats and ops.packed_active():
                img_out = ops.PackedImage(x.shape[0], 1, x.shape[1], x.shape[2], x.shape[3], x.device)
            # the LayerNorm between the block's two convolutions, and a block output whose only reader is the convolution
            # behind `out_image`, are consumed through their packed image alone
            x = ln(x, 0.01, st, img_out.slot(0) if img_out is not None else None, keep_f32=img_out is None, sole=True)
            img_in = img_out
        return x


class CRN_smaller(nn.Module):
    def __init__(self, input_channel=6, fg=False):
        super().__init__()
        ic = input_channel
        self.conv1_encoder = ConvBlock(2, ic, 64, (3, 3), 1)
        self.conv2_encoder = ConvBlock(2, 64, 128, (3, 3), 1)
        self.conv3_encoder = ConvBlock(2, 128, 128, (3, 3), 1)
        self.conv4_encoder = ConvBlock(2, 128, 256, (3, 3), 1)
        self.conv5_encoder = ConvBlock(2, 256, 256, (3, 3), 1)
        self.conv6_encoder = ConvBlock(2, 256, 512, (3, 3), 1)
        self.conv6_decoder = ConvBlock(2, ic + 512, 512, (3, 3), 1)
        self.conv5_decoder = ConvBlock(2, ic + 512 + 256, 512, (3, 3), 1)
        self.conv4_decoder = ConvBlock(2, ic + 512 + 256, 512, (3, 3), 1)
        self.conv3_decoder = ConvBlock(2, ic + 512 + 128, 512, (3, 3), 1)
        self.conv2_decoder = ConvBlock(2, ic + 512 + 128, 512, (3, 3), 1)
        self.conv1_decoder = ConvBlock(2, ic + 512 + 64, 256, (3, 3), 1)
        self.decoder = ConvBlock(2, ic + 256, 256, (3, 3), 1)
        self.out_conv = _Conv2d(256, 3, 1)
        self.fg = fg
        if fg:
            self.fg_conv = _Conv2d(256, 1, 1)

    def forward(self, label, sp):
        label = label.contiguous()
        pool = lambda t: ops.avg_pool(t, 3, 2, 1)
        down = lambda s: ops.resize(label, (s, s), align_corners=True)
        up = lambda t, s: ops.resize(t, (s, s), align_corners=True, lazy=True)      # only ever a source of the next decoder conv
        pool1 = pool(self.conv1_encoder(label))
        pool2 = pool(self.conv2_encoder(pool1))
        pool3 = pool(self.conv3_encoder(pool2))
        pool4 = pool(self.conv4_encoder(pool3))
        pool5 = pool(self.conv5_encoder(pool4))
        pool6 = pool(self.conv6_encoder(pool5))
        net_6 = up(self.conv6_decoder([down(sp // 64), pool6]), sp // 32)
        net_5 = up(self.conv5_decoder([down(sp // 32), pool5, net_6]), sp // 16)
        net_4 = up(self.conv4_decoder([down(sp // 16), pool4, net_5]), sp // 8)
        net_3 = up(self.conv3_decoder([down(sp // 8), pool3, net_4]), sp // 4)
        net_2 = up(self.conv2_decoder([down(sp // 4), pool2, net_3]), sp // 2)
        net_1 = up(self.conv1_decoder([down(sp // 2), pool1, net_2]), sp)
        head_img = ops.PackedImage(label.shape[0], 1, 256, sp, sp, label.device) if ops.packed_active() else None
        net = self.decoder([label, net_1], out_image=head_img)
        if self.fg:
            # the rgb head (256 -> 3) and the mask head (256 -> 1, sigmoid) read the same 256-channel tensor: one 1x1
            # convolution with the four output rows stacked packs `net` once and sends ONE gradient back to it
            # (separately: two packs of a 537 MB tensor, two data gradients and their 537 MB sum, per step at B=8)
            w = torch.cat([self.out_conv.weight, self.fg_conv.weight], 0)
            b = torch.cat([self.out_conv.bias, self.fg_conv.bias], 0)
            y = ops.conv2d(net, w, b, stride=1, pad=0, act=ACT_NONE, prepacked=head_img)
            return y[:, :3].contiguous(), torch.sigmoid(y[:, 3:4]).contiguous()
        return ops.conv2d(net, self.out_conv.weight, self.out_conv.bias, stride=1, pad=0, act=ACT_NONE, prepacked=head_img)
