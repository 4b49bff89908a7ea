"""Stage-4 generator / discriminator / loss modules on the HIP kernels.

Same class names, constructor arguments, forward signatures and state_dict keys as the live
surface of the reference's src/networks.py, so a checkpoint saved by
train/4.convLSTM_flowpro_interval.py:515-536 loads with ``load_state_dict`` and the scripts'
call sites drop on.  Internally the 24 body-part networks are ONE grouped network: every layer
holds a single [24*Cout, Cin, k, k] parameter and runs as one grouped launch
(``forward_grouped``); the reference loops 24 Python modules (src/networks.py:1657-1660,1821-1826).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import List, Sequence

import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID

NPARTS = 24
ENC_NC = [12, 24, 24, 24, 24, 48, 48, 96, 96]          # src/networks.py:1602
ENC_K = [5, 3, 3, 3, 3, 3, 3, 3, 3]                     # enc1 is 5x5 pad 2 (:1294)
ENC_S = [1, 2, 1, 2, 1, 2, 1, 2, 1]
_DEC_PACKED_OUT = True      # dec4 writes the output convolution's packed image
_VGG_SKIP_F32 = True        # untapped VGG conv -> conv layers write no fp32 output
SIZES = [200, 100, 50, 25, 13]


def _conv_init_(w: torch.Tensor, b, fan_in: int):
    bound = 1.0 / math.sqrt(fan_in)
    nn.init.uniform_(w, -bound, bound)
    if b is not None:
        nn.init.uniform_(b, -bound, bound)


class _GroupedStateDict(nn.Module):
    """Parameters are stored grouped ([24*Cout, ...]); state_dict()/load_state_dict() speak the
    reference's per-part keys through `self._key_map`: {param_name: key_template with {p}}."""

    def _init_grouped(self):
        self._key_map = OrderedDict()
        self._register_state_dict_hook(_GroupedStateDict._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    def _add(self, name: str, template: str, tensor: torch.Tensor):
        self.register_parameter(name, nn.Parameter(tensor))
        self._key_map[name] = template

    @staticmethod
    def _sd_hook(module, state_dict, prefix, local_metadata):
        for name, template in module._key_map.items():
            t = state_dict.pop(prefix + name)
            per = t.shape[0] // NPARTS
            for p in range(NPARTS):
                state_dict[prefix + template.format(p=p)] = t[p * per:(p + 1) * per]
        # reference key order: all keys of part 0, then part 1, ... (ModuleList order)
        def part_of(k):
            s = k[len(prefix):].split(".")
            return (0 if s[0].startswith("Downsampler") else 1, int(s[1]))
        items = sorted(((k, v) for k, v in state_dict.items() if k.startswith(prefix)),
                       key=lambda kv: part_of(kv[0]))
        for k, _ in items:
            state_dict.pop(k)
        for k, v in items:
            state_dict[k] = v
        return state_dict

    def _load_hook(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for name, template in self._key_map.items():
            keys = [prefix + template.format(p=p) for p in range(NPARTS)]
            if all(k in state_dict for k in keys):
                state_dict[prefix + name] = torch.cat([state_dict.pop(k) for k in keys], 0)


def _lrelu_conv(x_srcs, w, b, stride=1, pad=1, shared=None, prepacked=None, dst=None, keep_f32=True):
    return ops.conv2d(x_srcs, w, b, stride=stride, pad=pad, act=ACT_LRELU, slope=0.2, groups=NPARTS, shared=shared,
                      prepacked=prepacked, dst=dst, keep_f32=keep_f32)


def _dec_out_image(i: int, skip: torch.Tensor, c: int):
    """The LAST decoder layer (dec4) feeds only the 3-channel output convolution: on the packed bf16 path it writes that
    convolution's packed input image itself and no fp32 tensor (no packing pass in front of the output convolution, and its data
    gradient comes back as a packed dz: ops.mark_single_consumer).  None for the other layers (their consumer is a resize)."""
    if i != 3 or not _DEC_PACKED_OUT or not ops.packed_active():
        return None
    return ops.PackedImage(skip.shape[0], NPARTS, c, skip.shape[2], skip.shape[3], skip.device)


class _PartEncoderMixin:
    """enc1..enc9 of Downsampler_stack / Downsampler_convLSTM (src/networks.py:1124-1132,1294-1302)."""

    def _make_encoder(self):
        cin = 3
        for i, (c, k) in enumerate(zip(ENC_NC, ENC_K)):
            w = torch.empty(NPARTS * c, cin, k, k)
            b = torch.empty(NPARTS * c)
            _conv_init_(w, b, cin * k * k)
            self._add("enc%d_w" % (i + 1), "Downsampler_list.{p}.enc%d.enconv.0.weight" % (i + 1), w)
            self._add("enc%d_b" % (i + 1), "Downsampler_list.{p}.enc%d.enconv.0.bias" % (i + 1), b)
            cin = c

    def _encode(self, x, tap=None, skips_feed_one_conv=False, x_image=None):
        """enc1..enc9 -> [x1, x3, x5, x7, x9].  On the packed bf16 path every layer writes its output straight into the
        next layer's packed input image (ops.PackedImage): no separate packing pass between the convolutions.
        `tap(level, x_level, image)` (accumulate: the ConvLSTM of that level) is called as soon as a skip feature exists;
        it receives the image the feature was written into -- for the accumulate network that image is the ConvLSTM's
        [x, h] sequence image, whose x half enc_{i+1} then reads in place -- and returns what to collect for the level."""
        feats = []
        packed = ops.packed_active()
        img_in = x_image if packed else None       # enc1's packed input, when the caller made it (ops.atlas_to_parts_packed)
        for i in range(9):
            k, s = ENC_K[i], ENC_S[i]
            c = ENC_NC[i]
            N, H, W = x.shape[0], x.shape[2], x.shape[3]
            OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
            img_out, dst = None, None
            if packed and (i < 8 or tap is not None):
                # skip features of the accumulate network share the ConvLSTM's image (2c channels: x then h)
                lstm_img = tap is not None and i % 2 == 0
                img_out = ops.PackedImage(N, NPARTS, 2 * c if lstm_img else c, OH, OW, x.device)
                if lstm_img and c % 8:
                    # x and h share a plane (12 channels): the h quarter of step 0's images is read (against zero weights)
                    # before anything writes it
                    img_out.images(0, N // tap.T).buf.zero_()
                dst = img_out.slot(0, 0, pad_tail=not lstm_img)
            # fp32 copy of the output only where something still reads it: in the inpainter, the decoder's skip / resize inputs.
            # (Until round 4 also in front of the stride-2 layers wider than 16 channels, whose weight gradient once ran on the
            # fp32-input kernel; it has long read the packed image, and those copies -- 737 + 184 + 92 MB per step in the accumulate
            # net alone -- were written for nobody: -0.35 ms per step.)
            keep = tap is None and i % 2 == 0
            x = _lrelu_conv(x, getattr(self, "enc%d_w" % (i + 1)), getattr(self, "enc%d_b" % (i + 1)), stride=s, pad=k // 2,
                            prepacked=img_in, dst=dst, keep_f32=keep)
            if i % 2 == 1:
                ops.mark_single_consumer(x)          # x2, x4, x6, x8 feed enc_{i+1} only: dz handed over in bf16 (ops._fusable_producer)
            if i % 2 == 0:
                if (tap is not None or skips_feed_one_conv) and i < 8:
                    # read by enc_{i+1} and by exactly one more convolution / ConvLSTM (the level's ConvLSTM; the
                    # inpainter's decoder): their two data gradients meet inside the second kernel (ops.GradSlot)
                    ops.mark_two_consumers(x)
                feats.append(x if tap is None else tap(i // 2, x, img_out))
            img_in = img_out
        return feats     # x1, x3, x5, x7, x9 (or what `tap` made of them)


def _as_grouped(parts: Sequence[torch.Tensor]) -> torch.Tensor:
    """list[24] of (B,3,H,W) -> [B,72,H,W]; free when the list is already channel slices of one tensor."""
    t0 = parts[0]
    base = getattr(t0, "_base", None)
    if base is not None and base.dim() == 4 and base.shape[1] == 3 * len(parts) and base.is_contiguous():
        ok = all(getattr(p, "_base", None) is base and p.data_ptr() == base.data_ptr() + 4 * 3 * i * base.stride(1)
                 for i, p in enumerate(parts))
        if ok:
            return base
    return torch.cat(list(parts), dim=1)


def _as_list(t: torch.Tensor) -> List[torch.Tensor]:
    return [t[:, 3 * p:3 * p + 3] for p in range(NPARTS)]


class Accumulate_LSTM_no_loss(_GroupedStateDict, _PartEncoderMixin):
    """src/networks.py:1641-1662: 24 x (Downsampler_convLSTM -> Upsampler_stack_noEmbed)."""

    DEC_NC = [48, 24, 12, 6]

    def __init__(self):
        super().__init__()
        self._init_grouped()
        self._make_encoder()
        for i, c in enumerate([12, 24, 24, 48, 96]):          # convLSTM1..5 (:1304-1313)
            w = torch.empty(NPARTS * 4 * c, 2 * c, 3, 3)
            b = torch.empty(NPARTS * 4 * c)
            _conv_init_(w, b, 2 * c * 9)
            self._add("lstm%d_w" % (i + 1), "Downsampler_list.{p}.convLSTM%d.cell_list.0.conv.weight" % (i + 1), w)
            self._add("lstm%d_b" % (i + 1), "Downsampler_list.{p}.convLSTM%d.cell_list.0.conv.bias" % (i + 1), b)
        dec_in = [96 + 48, 24 + 48, 24 + 24, 12 + 12]          # (:1201-1204)
        for i, (ci, co) in enumerate(zip(dec_in, self.DEC_NC)):
            w = torch.empty(NPARTS * co, ci, 3, 3)
            b = torch.empty(NPARTS * co)
            _conv_init_(w, b, ci * 9)
            self._add("dec%d_w" % (i + 1), "Upsampler_list.{p}.dec%d.myconv.0.weight" % (i + 1), w)
            self._add("dec%d_b" % (i + 1), "Upsampler_list.{p}.dec%d.myconv.0.bias" % (i + 1), b)
        w = torch.empty(NPARTS * 3, 6, 3, 3)
        b = torch.empty(NPARTS * 3)
        _conv_init_(w, b, 6 * 9)
        self._add("out_w", "Upsampler_list.{p}.conv.weight", w)
        self._add("out_b", "Upsampler_list.{p}.conv.bias", b)

    def forward_grouped(self, x: torch.Tensor, T: int, x_image=None) -> torch.Tensor:
        """x: [T*B, 72, 200, 200] with image index t*B + b  ->  [B, 72, 200, 200].  `x_image`: the packed image of x
        (ops.atlas_to_parts_packed; x is then only its shape)."""
        TB = x.shape[0]
        B = TB // T

        def lstm(level, f, image):
            # runs right after the encoder layer that made f: the ConvLSTM only needs that level, and on the packed path it
            # completes the [x, h] image before enc_{i+1} reads x out of it
            seq = ops.share_gradslot(f, f.view(T, B, f.shape[1], f.shape[2], f.shape[3]))
            h, _ = ops.convlstm(seq, getattr(self, "lstm%d_w" % (level + 1)), getattr(self, "lstm%d_b" % (level + 1)),
                                groups=NPARTS, return_all=False, seq_image=image, return_state=False)
            return h
        lstm.T = T
        hs = self._encode(x, tap=lstm, x_image=x_image)
        x = hs[4]
        img = None
        for i in range(4):            # Upsampler_SE: bilinear(AC=True) to a fixed size, cat skip, conv+lrelu
            skip = hs[3 - i]
            up = ops.resize(x, (skip.shape[2], skip.shape[3]), align_corners=True, lazy=True)    # sampled while dec packs its input
            img = _dec_out_image(i, skip, self.DEC_NC[i])
            x = _lrelu_conv([up, skip], getattr(self, "dec%d_w" % (i + 1)), getattr(self, "dec%d_b" % (i + 1)),
                            dst=img.slot(0, 0, pad_tail=True) if img is not None else None, keep_f32=img is None)
        if img is not None:
            ops.mark_single_consumer(x)
        return ops.conv2d(x, self.out_w, self.out_b, stride=1, pad=1, act=ACT_NONE, groups=NPARTS, prepacked=img)

    def forward(self, x_in):
        """x_in: list[24][T] of (B,3,200,200) -> list[24] of (B,3,200,200)."""
        T = len(x_in[0])
        # image index t*B+b, part-major channels: the layout torch.cat(x, dim=0) gives per part (:1317)
        x = torch.cat([torch.cat([x_in[p][t] for p in range(NPARTS)], dim=1) for t in range(T)], dim=0)
        return _as_list(self.forward_grouped(x, T))


class Accumulate_LSTM(Accumulate_LSTM_no_loss):
    """src/networks.py:1593-1639: stage-1 variant with the atlas paste and masked L1 over the targets."""

    def forward(self, x_in, src_texture_mask, tgt_texture_mask, tgt_texture_im):
        T = len(x_in[0])
        x = torch.cat([torch.cat([x_in[p][t] for p in range(NPARTS)], dim=1) for t in range(T)], dim=0)
        return self._paste_and_loss(self.forward_grouped(x, T), src_texture_mask, tgt_texture_mask, tgt_texture_im)

    def forward_atlas(self, src_texture_im, src_texture_mask, tgt_texture_mask, tgt_texture_im):
        """Same as forward() with the reference textures given as atlases [B,T,3,800,1200] (the 24-part slicing of
        train/1...py:151-158 runs in one kernel)."""
        T = src_texture_im.shape[1]
        return self._paste_and_loss(self.forward_grouped(ops.atlas_to_parts(src_texture_im.contiguous()), T),
                                    src_texture_mask, tgt_texture_mask, tgt_texture_im)

    def _paste_and_loss(self, out, src_texture_mask, tgt_texture_mask, tgt_texture_im):
        """out [B,72,200,200] -> (atlas, loss), src/networks.py:1614-1639.  Masks are uint8 {0,1} with a singleton or
        3-wide channel axis ((B,T,1|3,800,1200)); the loss always compares with target 0 (`tgt_texture_im[:,0]`, :1634)."""
        B = out.shape[0]
        # atlas paste (:1614-1620): part p -> rows (p//6)*200, cols (p%6)*200
        atlas = out.view(B, 4, 6, 3, 200, 200).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, 800, 1200)
        common = torch.zeros_like(src_texture_mask[:, 0])
        for i in range(src_texture_mask.shape[1]):
            common = common | src_texture_mask[:, i]
        loss = None
        real = tgt_texture_im[:, 0].float().contiguous()
        atlas_c = atlas.contiguous()
        for i in range(tgt_texture_mask.shape[1]):
            area = (common & tgt_texture_mask[:, i]).float().contiguous()
            gen = ops.mul_bcast(atlas_c, area)
            rl = ops.mul_bcast(real, area)
            term = ops.l1_loss(gen, rl, 1.0)
            loss = term if loss is None else loss + term
        return atlas, loss.squeeze(0)


class UNet_inpainter(_GroupedStateDict, _PartEncoderMixin):
    """src/networks.py:1805-1828: 24 encoders -> 72-channel global embed -> 24 decoders."""

    DEC_NC = [96, 48, 24, 12]

    def __init__(self):
        super().__init__()
        self._init_grouped()
        self._make_encoder()
        w = torch.empty(NPARTS * 3, 96, 3, 3)
        b = torch.empty(NPARTS * 3)
        _conv_init_(w, b, 96 * 9)
        self._add("cmp_w", "Downsampler_list.{p}.enc_compress.enconv.0.weight", w)
        self._add("cmp_b", "Downsampler_list.{p}.enc_compress.enconv.0.bias", b)
        dec_in = [96 + 48 + 72, 24 + 96, 24 + 48, 12 + 24]      # (:1156-1159)
        for i, (ci, co) in enumerate(zip(dec_in, self.DEC_NC)):
            w = torch.empty(NPARTS * co, ci, 3, 3)
            b = torch.empty(NPARTS * co)
            _conv_init_(w, b, ci * 9)
            self._add("dec%d_w" % (i + 1), "Upsampler_list.{p}.dec%d.myconv.0.weight" % (i + 1), w)
            self._add("dec%d_b" % (i + 1), "Upsampler_list.{p}.dec%d.myconv.0.bias" % (i + 1), b)
        w = torch.empty(NPARTS * 3, 12, 3, 3)
        b = torch.empty(NPARTS * 3)
        _conv_init_(w, b, 12 * 9)
        self._add("out_w", "Upsampler_list.{p}.conv.weight", w)
        self._add("out_b", "Upsampler_list.{p}.conv.bias", b)

    def forward_grouped(self, x: torch.Tensor) -> torch.Tensor:
        """x: [B, 72, 200, 200] -> [B, 72, 200, 200]."""
        feats = self._encode(x, skips_feed_one_conv=True)     # x1..x7: enc_{i+1} and the decoder's skip input below
        embed = _lrelu_conv(feats[4], self.cmp_w, self.cmp_b)          # [B, 72, 13, 13] == cat of the 24 embeds (:1824)
        # dec1 input = up(cat[x9, global_embed]) ++ x7 ; bilinear is per channel so the cat is never built
        skip = feats[3]
        size = (skip.shape[2], skip.shape[3])
        up9 = ops.resize(feats[4], size, align_corners=True, lazy=True)
        upe = ops.resize(embed, size, align_corners=True, lazy=True)
        x = _lrelu_conv([up9, upe, skip], self.dec1_w, self.dec1_b, shared=[False, True, False])
        img = None
        for i in range(1, 4):
            skip = feats[3 - i]
            up = ops.resize(x, (skip.shape[2], skip.shape[3]), align_corners=True, lazy=True)
            img = _dec_out_image(i, skip, self.DEC_NC[i])
            x = _lrelu_conv([up, skip], getattr(self, "dec%d_w" % (i + 1)), getattr(self, "dec%d_b" % (i + 1)),
                            dst=img.slot(0, 0, pad_tail=True) if img is not None else None, keep_f32=img is None)
        if img is not None:
            ops.mark_single_consumer(x)
        return ops.conv2d(x, self.out_w, self.out_b, stride=1, pad=1, act=ACT_NONE, groups=NPARTS, prepacked=img)

    def forward(self, texture_list):
        return _as_list(self.forward_grouped(_as_grouped(texture_list).contiguous()))


# ------------------------------------------------------------------------------------------------
# single-part building blocks kept for API parity (src/networks.py:868-878, 896-909)
# ------------------------------------------------------------------------------------------------
class _Conv(nn.Module):
    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        _conv_init_(self.weight, self.bias, cin * k * k)


class Downsampler(nn.Module):
    def __init__(self, input_nc, output_nc, kernel_size=3, stride=1, padding=1):
        super().__init__()
        self.enconv = nn.Sequential(_Conv(input_nc, output_nc, kernel_size), nn.Identity())
        self.stride, self.padding = stride, padding

    def forward(self, x):
        c = self.enconv[0]
        return ops.conv2d(x, c.weight, c.bias, stride=self.stride, pad=self.padding, act=ACT_LRELU, slope=0.2)


class Upsampler_SE(nn.Module):
    def __init__(self, input_nc, output_nc, kernel_size=3, padding=1, output_size=50):
        super().__init__()
        self.myconv = nn.Sequential(_Conv(input_nc, output_nc, kernel_size), nn.Identity())
        self.output_size, self.padding = output_size, padding

    def forward(self, x, enc_x):
        up = ops.resize(x, (self.output_size, self.output_size), align_corners=True)
        c = self.myconv[0]
        return ops.conv2d([up, enc_x], c.weight, c.bias, stride=1, pad=self.padding, act=ACT_LRELU, slope=0.2)


# ------------------------------------------------------------------------------------------------
# discriminators (src/networks.py:356-456)
# ------------------------------------------------------------------------------------------------
class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    _nbt_pending = 0        # train-mode calls not yet added to the num_batches_tracked buffer

    def apply_bn(self, x, act, slope, training, residual=None, batch_parts=1):
        if training:
            # The counter is bookkeeping (momentum is fixed at 0.1, nothing on the device reads it): counted on the host and added to
            # the buffer when the state is read (55 one-element launches per step otherwise).  A hipGraph capture records the
            # device-side increment instead, since the Python code does not run again on replay.
            # Contract: the buffer is current whenever state_dict() is taken and, on the trainers, between steps
            # (step.flush_bn_counters: one batched add per step); inside a step it lags by the calls made so far.
            if x.is_cuda and torch.cuda.is_current_stream_capturing():
                self.num_batches_tracked += batch_parts
            else:
                self._nbt_pending += batch_parts
        return ops.batchnorm_act(x, self.weight, self.bias, self.running_mean, self.running_var, training, act,
                                 slope, residual, batch_parts=batch_parts)


def _bn_flush(self):
    if self._nbt_pending:
        self.num_batches_tracked += self._nbt_pending
        self._nbt_pending = 0


def _bn_save(self, destination, prefix, keep_vars):
    _bn_flush(self)
    nn.Module._save_to_state_dict(self, destination, prefix, keep_vars)


def _bn_load(self, *args, **kwargs):
    self._nbt_pending = 0
    return nn.Module._load_from_state_dict(self, *args, **kwargs)


_BN.flush_counters = _bn_flush
_BN._save_to_state_dict = _bn_save
_BN._load_from_state_dict = _bn_load


class _Linear(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.empty(o))
        _conv_init_(self.weight, self.bias, i)


class _DCGANDiscriminator(nn.Module):
    def _build(self, chans: Sequence[int], input_channel: int, feat: int):
        layers: List[nn.Module] = []
        cin = input_channel
        self._plan = []                                   # (conv idx, bn idx or None)
        for i, c in enumerate(chans):
            conv_idx = len(layers)
            layers.append(_Conv(cin, c, 3, bias=False))
            bn_idx = None
            if i > 0:
                bn_idx = len(layers)
                layers.append(_BN(c))
            layers.append(nn.Identity())                  # the LeakyReLU slot keeps the reference indices
            self._plan.append((conv_idx, bn_idx))
            cin = c
        self.main = nn.Sequential(*layers)
        self.classifier = nn.Sequential(_Linear(feat, 100), nn.Identity(), _Linear(100, 1), nn.Identity())

    def forward(self, input, batch_parts: int = 1):
        """`batch_parts` = k: the batch holds k equal chunks that the reference passes through the network in k separate calls
        (real images, generated images: train/4...py:380-394).  Convolutions and the classifier see them as one batch (they
        are per-sample), every BatchNorm normalises each chunk with its own statistics and updates the running statistics
        chunk after chunk -- the result of the k calls, with half the kernel launches."""
        x = input
        for conv_idx, bn_idx in self._plan:
            conv = self.main[conv_idx]
            if bn_idx is None:
                x = ops.conv2d(x, conv.weight, None, stride=2, pad=1, act=ACT_LRELU, slope=0.2)
            else:
                x = ops.conv2d(x, conv.weight, None, stride=2, pad=1, act=ACT_NONE)
                x = self.main[bn_idx].apply_bn(x, ACT_LRELU, 0.2, self.training, batch_parts=batch_parts)
        x = x.reshape(x.size(0), -1)
        l0, l2 = self.classifier[0], self.classifier[2]
        x = ops.linear(x, l0.weight, l0.bias, ACT_LRELU, 0.2)
        return ops.linear(x, l2.weight, l2.bias, ACT_SIGMOID, 0.0)


class ImageDiscriminator(_DCGANDiscriminator):
    def __init__(self, ndf, input_channel=3):
        super().__init__()
        self._build([ndf, ndf * 2, ndf * 2, ndf * 4, ndf * 4, ndf * 8], input_channel, ndf * 8 * 4 * 4)


class FaceDiscriminator(_DCGANDiscriminator):
    def __init__(self, ndf, input_channel=3):
        super().__init__()
        self._build([ndf, ndf * 2, ndf * 2, ndf * 4], input_channel, ndf * 4 * 4 * 4)


# ------------------------------------------------------------------------------------------------
# VGG perceptual + L1 loss (src/networks.py:70-125)
# ------------------------------------------------------------------------------------------------
_VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def vgg_preprocess(x):
    return ops.vgg_preprocess(x)


class VGG19_CRN(nn.Module):
    """VGG19 `features` with AvgPool in place of MaxPool (:76-78).  torchvision's in-place ReLU
    aliases the tapped tensors, so the five taps (layers 2,7,12,21,30) are POST-ReLU;
    conv5_3/5_4/pool5 are computed by the reference and never used -- skipped here.
    Weights: the reference downloads ImageNet weights (torchvision 0.4.0); no copy exists offline,
    so they are whatever load_state_dict / the default init provides ("parity unpinned" vs the
    pretrained net)."""

    TAPS = (2, 7, 12, 21, 30)

    def __init__(self, requires_grad=False):
        super().__init__()
        self.vgg_model = nn.Module()
        self._layers = []
        idx, cin = 0, 3
        for v in _VGG19_CFG:
            if v == "M":
                self._layers.append(("pool", idx))
                idx += 1
            else:
                conv = _Conv(cin, v, 3)
                self.vgg_model.add_module(str(idx), conv)
                self._layers.append(("conv", idx))
                idx += 2
                cin = v
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        feats = []
        layers = [l for l in self._layers if l[1] <= self.TAPS[-1]]
        img_in = None
        for li, (kind, idx) in enumerate(layers):
            if kind == "pool":
                x = ops.avg_pool(x, 2, 2, 0)
                img_in = None
            else:
                conv = getattr(self.vgg_model, str(idx))
                # conv -> conv edges: the epilogue writes the next layer's packed input image (packed bf16 path only)
                nxt_is_conv = li + 1 < len(layers) and layers[li + 1][0] == "conv"
                img_out = None
                if nxt_is_conv and ops.packed_active():
                    img_out = ops.PackedImage(x.shape[0], 1, conv.weight.shape[0], x.shape[2], x.shape[3], x.device)
                # ... and an untapped layer's fp32 output has no reader at all (the next convolution reads the image, the backward pass
                # takes the ReLU's sign from it): 7 layers, 307 MB per pass over 8 frames, not written
                x = ops.conv2d(x, conv.weight, conv.bias, stride=1, pad=1, act=ACT_RELU, prepacked=img_in,
                               dst=img_out.slot(0, 0, pad_tail=True) if img_out is not None else None,
                               keep_f32=img_out is None or idx in self.TAPS or not _VGG_SKIP_F32)
                img_in = img_out
                if idx in self.TAPS:
                    feats.append(x)
                elif nxt_is_conv:
                    ops.mark_single_consumer(x)      # read by the next convolution only (the tapped layers also feed the loss)
        return feats


class VGGLoss_CRN(nn.Module):
    def __init__(self, weights=(1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)):
        super().__init__()
        self.vgg = VGG19_CRN()
        self.weights = list(weights)

    def forward(self, x, y, y_vgg=None):
        """y_vgg: features of y computed earlier by the same (frozen) network, e.g. on a side stream."""
        x_vgg = self.vgg(x)
        if y_vgg is None:
            with torch.no_grad():
                y_vgg = self.vgg(y)
        loss = None
        for i in range(len(x_vgg)):
            term = ops.l1_loss(x_vgg[i], y_vgg[i], self.weights[i])
            loss = term if loss is None else loss + term
        return loss


class VGG_l1_loss(nn.Module):
    def __init__(self):
        super().__init__()
        self.vgg_loss = VGGLoss_CRN(weights=[1 / 2.6, 1 / 4.8, 1 / 3.7, 1 / 5.6, 10 / 1.5])

    @torch.no_grad()
    def target_features(self, y):
        """Everything of the loss that depends on the target alone (the network is frozen): the preprocessed
        target and its five feature maps.  Pass the result as `forward(x, y, target=...)`."""
        yp = vgg_preprocess(y.contiguous())
        return yp, self.vgg_loss.vgg(yp)

    def forward(self, x, y, target=None):
        xp = vgg_preprocess(x.contiguous())
        if target is None:
            yp, y_vgg = vgg_preprocess(y.contiguous()), None
        else:
            yp, y_vgg = target
        loss = self.vgg_loss(xp, yp, y_vgg) + ops.l1_loss(xp, yp, 1.0)
        return loss.squeeze(0)


# ------------------------------------------------------------------------------------------------
# texture warp (train/4.convLSTM_flowpro_interval.py:43-76 == src/networks.py:36-68)
# ------------------------------------------------------------------------------------------------
def texture_warp_pytorch(tex_parts, IUV, device=None, align_corners=False):
    """tex_parts: list[24] of (3,200,200); IUV: (S,S,3) uint8 ndarray / tensor -> (3,S,S).
    The batched form is ops.texture_warp([B,72,200,200], [B,S,S,3] uint8)."""
    tex = torch.cat([t.float() for t in tex_parts], dim=0).unsqueeze(0).contiguous()
    if not torch.is_tensor(IUV):
        IUV = torch.from_numpy(IUV)
    iuv = IUV.to(device=tex.device, dtype=torch.uint8).unsqueeze(0).contiguous()
    return ops.texture_warp(tex, iuv, align_corners).squeeze(0)
