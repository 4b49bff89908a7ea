"""torch.autograd bindings of the C-ABI kernels (include/jafpro_hip.h).

PyTorch supplies device memory, the current HIP stream and the autograd tape; every number is
produced by libjafpro_hip.so.  All ops require contiguous fp32 GPU tensors and raise
RuntimeError otherwise (the reference extension does the same through CHECK_CUDA /
CHECK_CONTIGUOUS, rasterize_cuda.cpp:66-68).
"""
from __future__ import annotations

import collections
import ctypes
import math
import os
import weakref
from typing import List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function

from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, PACK_DGRAD, PACK_DGRAD_LSTM, PACK_FWD,
                   PACK_LSTM, PREC_BF16, PREC_BF16X3, PREC_F32, ConvDesc, ConvPlan, PackedIO, check, lib)

__all__ = ["set_precision", "get_precision", "conv2d", "convlstm", "layernorm_lrelu", "batchnorm_act", "avg_pool", "resize",
           "reflect_pad", "texture_warp", "grid_sample", "blend", "mul_bcast", "part_mask_mul",
           "atlas_to_parts", "atlas_to_parts_packed", "vgg_preprocess", "l1_loss", "bce_loss", "linear", "adam_step",
           "project_faces", "rasterize_fim_wim", "bc_transform", "axpby", "conv_transpose2d", "ACT_NONE", "ACT_LRELU",
           "ACT_RELU", "ACT_SIGMOID", "ACT_TANH"]


# --------------------------------------------------------------------------------------------
# plumbing
# --------------------------------------------------------------------------------------------
def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _s():
    """torch's current HIP stream of the current device as a C handle.  The raw accessors cost ~0.3 us; the public
    `torch.cuda.current_stream()` builds a Stream object (~9 us) -- 10 ms of host time per train step at ~1100 launches."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a GPU tensor (jafpro_amd has no CPU path)" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


_GRID_Z = 65535


def _n_chunks(N: int, planes_per_image: int):
    """(n0, n1) image ranges whose plane count fits the z extent of a HIP grid: the per-plane kernels put
    (image, channel) on blockIdx.z, so batches beyond 65535 planes go out as several launches instead of failing."""
    per = max(1, _GRID_Z // max(1, planes_per_image))
    return [(n0, min(N, n0 + per)) for n0 in range(0, N, per)]


# --------------------------------------------------------------------------------------------
# convolution core
# --------------------------------------------------------------------------------------------
_PLAN_CACHE = {}
_PACK_CACHE = {}
_PACK_CACHE_MAX = 4096
_PACK_KEYS_BY_ID = {}
_PROF = None
_PRECISION = PREC_F32
# The bf16 modes run on the packed-input (DMA-staged) kernels of csrc/conv_dma.hip / conv_dma_split.hip -- the only bf16 kernels
# there are since round 5 (the fp32-input staging kernels of rounds 1-2 were deleted); f32 runs on csrc/conv.hip.
_USE_PACKED = True
_PREC_NAMES = {"f32": PREC_F32, "bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "mixed": PREC_BF16X3}
# "mixed": the forward pass in split-bf16 (parity-grade: the generated frame and every loss <= 1e-3 of the fp32 oracle), the backward
# pass -- data and weight gradients -- in plain bf16.  The forward's saved operand images are split images; a backward launch reads
# their hi planes, which ARE the bf16 images (jaf_conv2d_wgrad_packed_ws_x, jaf_packed_io.dz_mask_split, jaf_conv2d_pack_dz_dt2).
_MIXED = False


def set_precision(name: str) -> str:
    """Matrix-core arithmetic of every convolution issued from now on (include/jafpro_hip.h
    JAF_PREC_*): "f32" exact fp32 MFMA (the <=1e-3 parity path), "bf16" bf16 operands / fp32
    accumulate (BASELINE configs[2]), "bf16x3" split-bf16 (hi+lo) emulation of fp32 products.
    Tensors stay fp32 in HBM in every mode.  Returns the previous setting."""
    global _PRECISION, _MIXED
    if name not in _PREC_NAMES:
        raise ValueError("precision must be one of %s" % sorted(_PREC_NAMES))
    prev = get_precision()
    _PRECISION = _PREC_NAMES[name]
    _MIXED = name == "mixed"
    return prev


def get_precision() -> str:
    if _MIXED:
        return "mixed"
    return {PREC_F32: "f32", PREC_BF16: "bf16", PREC_BF16X3: "bf16x3"}[_PRECISION]


def _bwd_mode():
    """(arithmetic, packed) a node created now runs its backward pass in: the forward's, except under "mixed"."""
    return (PREC_BF16 if _MIXED else _PRECISION, _USE_PACKED)


def _fwd_key():
    return (_PRECISION, _MIXED)


# bf16 STORAGE in the "bf16" arithmetic mode (BASELINE configs[2] names bf16; SURVEY 8(d) "bf16 storage / fp32 accumulate"): tensors
# that only kernels of this library read between two convolutions -- the ConvLSTM's cell state and its time-loop gradients, the
# pre-LayerNorm convolution outputs of the CRN and the gradients flowing back into them -- are kept in bf16 instead of fp32.  Every
# operand of every convolution is rounded to bf16 in that mode anyway; "f32" and "bf16x3" (the parity-grade modes) never use it.
_BF16_STORAGE = True


def set_bf16_storage(flag: bool) -> bool:
    """False: the "bf16" mode keeps every NCHW tensor in fp32 as in rounds 1-4 (A/B and tests).  Returns the previous setting."""
    global _BF16_STORAGE
    prev, _BF16_STORAGE = _BF16_STORAGE, bool(flag)
    return prev


def bf16_storage_active() -> bool:
    return _BF16_STORAGE and _PRECISION == PREC_BF16 and _PACKED_IMAGES


def bf16_handles_active() -> bool:
    """Storage-less autograd handles are made bf16 tensors -- so that the gradients autograd carries for them are produced and
    consumed in bf16 -- under bf16 storage, and in the "mixed" mode, whose backward pass is a bf16 pass (a handle holds no data:
    the parity-grade forward is untouched)."""
    return bf16_storage_active() or (_BF16_STORAGE and _MIXED and _PACKED_IMAGES)


class _arith:
    """`with _arith(ctx.mode):` -- a backward pass runs in the arithmetic its forward ran in, whatever
    set_precision says by then (its saved packed images / bf16 gates belong to that mode)."""
    __slots__ = ("mode", "prev")

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        global _PRECISION, _USE_PACKED, _MIXED
        self.prev = (_PRECISION, _USE_PACKED, _MIXED)
        _PRECISION, _USE_PACKED = self.mode
        _MIXED = False              # (inside a backward pass every launch is of the one arithmetic the node says)

    def __exit__(self, *exc):
        global _PRECISION, _USE_PACKED, _MIXED
        _PRECISION, _USE_PACKED, _MIXED = self.prev
        return False


_NAME_BUF = ctypes.create_string_buffer(160)


def _launched() -> str:
    """Name of the kernel instantiation the last convolution-family launch of this thread picked, as rocprofv3 prints it
    (jaf_last_kernel_name: written by the launch code itself while a profiler is set, see set_profiler)."""
    check(lib().jaf_last_kernel_name(_NAME_BUF, len(_NAME_BUF)), "jaf_last_kernel_name")
    return _NAME_BUF.value.decode()


class KernelProfiler:
    """Optional per-launch timing of the MFMA kernels with events on the launch stream
    (bench.py's roofline leg).  Records (kernel name, algorithmic FLOPs, start, end)."""

    def __init__(self):
        self.records = []

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def end(self, name, flops, ev0, nbytes=0.0):
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream())
        self.records.append((name, float(flops), ev0, ev1, float(nbytes)))

    def summary(self):
        """{kernel: {launches, ms, flops, bytes}}: MFMA kernels carry algorithmic FLOPs, the
        HBM-bound gather / blend / pack kernels carry algorithmic bytes."""
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1, nbytes in self.records:
            r = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            r["launches"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["flops"] += flops
            r["bytes"] += nbytes
        return out


def set_profiler(p):
    global _PROF
    _PROF = p
    lib().jaf_kernel_names(1 if p is not None else 0)


class _hbm:
    """with _hbm(kernel name, algorithmic bytes): <one launch>  -- timed only while a profiler is set."""
    __slots__ = ("name", "nbytes", "ev")

    def __init__(self, name, nbytes):
        self.name, self.nbytes = name, nbytes

    def __enter__(self):
        self.ev = _PROF.begin() if _PROF is not None else None

    def __exit__(self, *exc):
        if self.ev is not None and exc[0] is None:
            _PROF.end(self.name if self.name is not None else _launched(), 0.0, self.ev, self.nbytes)
        return False


_DESC_CACHE: dict = {}


def _make_desc(N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, specs, w_cin_tot,
               w_cin_off, out_ctot, out_coff, act, slope) -> ConvDesc:
    """The descriptor of one launch; descriptors are read-only on both sides of the ABI, so equal arguments share one
    object (filling the ctypes struct field by field was ~15 us per launch)."""
    key = (N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, tuple(specs), w_cin_tot, w_cin_off, out_ctot,
           out_coff, act, float(slope), _PRECISION)
    hit = _DESC_CACHE.get(key)
    if hit is not None:
        return hit
    d = _fill_desc(N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, specs, w_cin_tot, w_cin_off, out_ctot,
                   out_coff, act, slope)
    if len(_DESC_CACHE) > 4096:
        _DESC_CACHE.clear()
    _DESC_CACHE[key] = d
    return d


def _fill_desc(N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, specs, w_cin_tot,
               w_cin_off, out_ctot, out_coff, act, slope) -> ConvDesc:
    d = ConvDesc()
    d.N, d.G, d.Cin, d.Cout = N, G, Cin, Cout
    d.H, d.W, d.OH, d.OW = H, W, OH, OW
    d.KH, d.KW, d.stride = KH, KW, stride
    d.pad_t, d.pad_l, d.dil_in = pad_t, pad_l, dil
    d.nsrc = len(specs)
    for i, (c, ctot, coff, gs) in enumerate(specs):
        d.src_c[i], d.src_ctot[i], d.src_coff[i], d.src_gstride[i] = c, ctot, coff, gs
    d.w_cin_tot, d.w_cin_off = w_cin_tot, w_cin_off
    d.out_ctot, d.out_coff = out_ctot, out_coff
    d.act, d.slope = act, slope
    d.precision = _PRECISION
    return d


def _packed_prec(prec) -> bool:
    """JAF_PREC_BF16 and JAF_PREC_BF16X3 (csrc/conv_dma_split.hip: hi + lo operand images, three matrix-core instructions per
    product) run on the packed-input kernels."""
    return prec == PREC_BF16 or prec == PREC_BF16X3


def _packed_path_now() -> bool:
    return _packed_prec(_PRECISION)


def _packed_path(d: ConvDesc) -> bool:
    return _packed_prec(d.precision)


def _plan(key, d: ConvDesc, lstm: int, flags: int = 0) -> ConvPlan:
    packed = _packed_path(d)
    k = (key, lstm, d.precision, packed, flags if packed else 0)
    pl = _PLAN_CACHE.get(k)
    if pl is None:
        pl = ConvPlan()
        if packed:
            check(lib().jaf_conv2d_plan_packed_ex(ctypes.byref(d), lstm, flags, ctypes.byref(pl)), "jaf_conv2d_plan_packed_ex")
        else:
            check(lib().jaf_conv2d_plan(ctypes.byref(d), lstm, ctypes.byref(pl)), "jaf_conv2d_plan")
        _PLAN_CACHE[k] = pl
    return pl


def pack_input(srcs: Sequence[torch.Tensor], d: ConvDesc, lazy=None) -> torch.Tensor:
    """bf16 [N][G][ceil(Cin/8)][H][W][8] image of a descriptor's (concatenated, grouped) input: converted
    once, consumed by the forward conv and the weight gradient (or by dgrad and wgrad for a dz)."""
    nbytes = int(lib().jaf_conv2d_packed_input_bytes(ctypes.byref(d)))
    if nbytes <= 0:
        raise RuntimeError("jaf_conv2d_packed_input_bytes: invalid descriptor")
    xp = torch.empty(nbytes, device=srcs[0].device, dtype=torch.uint8)
    ng8 = (d.Cin + 7) // 8
    chunks = _n_chunks(d.N, d.G * ng8)
    if lazy is None:
        lazy = [getattr(t, "_jaf_lazy", None) for t in srcs]
    if any(l is not None for l in lazy):
        # up-sampled sources given at their low resolution: sampled while packing (jaf_conv2d_pack_input_resized)
        I3 = ctypes.c_int32 * 3
        sh, sw, al = I3(0, 0, 0), I3(0, 0, 0), I3(0, 0, 0)
        real = []
        for i, (t, l) in enumerate(zip(srcs, lazy)):
            if l is None:
                real.append(t)
            else:
                real.append(l[0])
                sh[i], sw[i], al[i] = int(l[0].shape[2]), int(l[0].shape[3]), 1 if l[1] else 0
        with _hbm(None, sum(4.0 * r.numel() for r in real) + nbytes):      # (None: the kernel the launch picks, see _launched)
            per = nbytes // d.N
            for n0, n1 in chunks:
                dc = type(d).from_buffer_copy(d)
                dc.N = n1 - n0
                ps = [_p(t[n0:n1]) for t in real] + [None] * (3 - len(real))
                check(lib().jaf_conv2d_pack_input_resized(_s(), ctypes.byref(dc), ps[0], ps[1], ps[2], sh, sw, al,
                                                          _p(xp[n0 * per:n1 * per])), "jaf_conv2d_pack_input_resized")
        return xp
    with _hbm("conv_pack_input_kernel", 4.0 * d.N * d.G * d.Cin * d.H * d.W + nbytes):
        if len(chunks) == 1:
            ps = [_p(t) for t in srcs] + [None] * (3 - len(srcs))
            check(lib().jaf_conv2d_pack_input(_s(), ctypes.byref(d), ps[0], ps[1], ps[2], _p(xp)), "jaf_conv2d_pack_input")
        else:       # more (image, group, plane) triples than a grid's z extent: a few launches over image ranges
            per = nbytes // d.N
            for n0, n1 in chunks:
                dc = type(d).from_buffer_copy(d)
                dc.N = n1 - n0
                ps = [_p(t[n0:n1]) for t in srcs] + [None] * (3 - len(srcs))
                check(lib().jaf_conv2d_pack_input(_s(), ctypes.byref(dc), ps[0], ps[1], ps[2], _p(xp[n0 * per:n1 * per])),
                      "jaf_conv2d_pack_input")
    return xp


class PackedImage:
    """A convolution's input as the kernels want it -- bf16 [N][G][ng8][H*W][8], `C` channels per group padded to a
    multiple of 8 -- allocated BEFORE its producers run: each producer (conv / ConvLSTM epilogue, LayerNorm, ...)
    writes its channels into its slot, so the concatenation the consumer reads (`cat[x, h]`, `cat[up, skip]`) is
    never built in fp32 and no separate packing pass exists (include/jafpro_hip.h, jaf_packed_io).  Only made while
    the packed bf16 path is active (`packed_active()`); module code passes None otherwise and everything falls back to
    jaf_conv2d_pack_input."""
    __slots__ = ("buf", "N", "G", "C", "ng8", "H", "W", "split")

    def __init__(self, N: int, G: int, C: int, H: int, W: int, device, zero: bool = False, buf: Optional[torch.Tensor] = None,
                 split: Optional[bool] = None):
        self.N, self.G, self.C, self.H, self.W = int(N), int(G), int(C), int(H), int(W)
        self.ng8 = (self.C + 7) // 8
        # split-bf16 (JAF_PREC_BF16X3): every channel group as a hi and a lo plane, [N][G][ng8][hi, lo][H*W][8] -- twice the bytes
        self.split = (_PRECISION == PREC_BF16X3) if split is None else bool(split)
        n = self.N * self.per_image
        if buf is None:
            buf = (torch.zeros if zero else torch.empty)(n, device=device, dtype=torch.uint8)
        self.buf = buf

    @property
    def per_image(self) -> int:
        """Bytes of one image (all groups and planes)."""
        return self.G * self.ng8 * self.H * self.W * 16 * (2 if self.split else 1)

    def images(self, n0: int, n: int) -> "PackedImage":
        """The sub-image holding images n0 .. n0+n (a view of the same memory)."""
        per = self.per_image
        return PackedImage(n, self.G, self.C, self.H, self.W, self.buf.device, buf=self.buf[n0 * per:(n0 + n) * per], split=self.split)

    def slot(self, coff: int = 0, img_off: int = 0, pad_tail: bool = False) -> "PackedDst":
        return PackedDst(self, coff, img_off, pad_tail)


class PackedDst:
    """Where a producer writes inside a PackedImage: channels [coff, coff + Cout) of every group of image n + img_off;
    `pad_tail`: this producer owns the image's last channels and also zeroes the padding up to the next multiple of 8."""
    __slots__ = ("image", "coff", "img_off", "pad_tail")

    def __init__(self, image: PackedImage, coff: int, img_off: int, pad_tail: bool):
        if coff % 4:
            raise ValueError("packed destination channel offset must be a multiple of 4")
        self.image, self.coff, self.img_off, self.pad_tail = image, int(coff), int(img_off), bool(pad_tail)


_PACKED_IMAGES = True


def set_packed_images(flag: bool) -> bool:
    """A/B switch: False sends every layer back through jaf_conv2d_pack_input (same numbers, one more pass per layer)."""
    global _PACKED_IMAGES
    prev, _PACKED_IMAGES = _PACKED_IMAGES, bool(flag)
    return prev


_Y_SIGN_FROM_IMAGE = True
_LAZY_RESIZE = True


def set_lazy_resize(flag: bool) -> bool:
    """A/B switch of ops.resize(lazy=True): False materialises every resize (jaf_resize_fwd) again."""
    global _LAZY_RESIZE
    prev, _LAZY_RESIZE = _LAZY_RESIZE, bool(flag)
    return prev


def lazy_resize_active() -> bool:
    # (split-bf16 too: jaf_conv2d_pack_input_resized writes hi + lo planes, and the weight gradient reads the packed image)
    return _LAZY_RESIZE and _packed_path_now()


def packed_active() -> bool:
    """True while convolutions run on the packed-input kernels (bf16, or split-bf16 with hi + lo planes): the modes
    PackedImages apply to."""
    return _PACKED_IMAGES and _packed_path_now()


def _io_struct(prepacked: Optional[PackedImage], dst: Optional[PackedDst], skip_f32: bool = False,
               accumulate: bool = False, out2: Optional[torch.Tensor] = None, split: int = 0, dz_fuse=None,
               out_bf16: bool = False, state_bf16: bool = False) -> Optional[PackedIO]:
    out2_bf16 = out2 is not None and out2.dtype == torch.bfloat16
    if prepacked is None and dst is None and not accumulate and out2 is None and not (out_bf16 or state_bf16):
        return None
    io = PackedIO()
    io.out_bf16, io.out2_bf16, io.state_bf16 = int(out_bf16), int(out2_bf16), int(state_bf16)
    if dz_fuse is not None:       # (mask image buffer, its planes per (image, group), channel offset, act' below zero, dbias or None)
        mbuf, mng8, mcoff, mslope, dbias = dz_fuse[:5]
        io.dz_mask, io.dz_mask_ng8, io.dz_mask_coff, io.dz_slope = mbuf.data_ptr(), int(mng8), int(mcoff), float(mslope)
        io.dz_mask_split = 1 if (len(dz_fuse) > 5 and dz_fuse[5] and _PRECISION == PREC_BF16) else 0
        io.dz_dbias = dbias.data_ptr() if dbias is not None else None
    io.accumulate_f32 = 1 if accumulate else 0
    if out2 is not None:
        io.out2, io.split_rows = out2.data_ptr(), split
    io.in_ng8_tot = prepacked.ng8 if prepacked is not None else 0
    if dst is not None:
        io.dst = dst.image.buf.data_ptr()
        io.dst_ng8_tot, io.dst_coff, io.dst_img_off, io.dst_pad_tail = dst.image.ng8, dst.coff, dst.img_off, 1 if dst.pad_tail else 0
    io.skip_f32 = 1 if skip_f32 else 0
    return io


def _check_image(img: PackedImage, N, G, Cin, H, W, what: str):
    if img.split != (_PRECISION == PREC_BF16X3):
        raise RuntimeError("%s: packed image made in another arithmetic mode (split = %s)" % (what, img.split))
    if (img.N, img.G, img.H, img.W) != (N, G, H, W) or img.C < Cin:
        raise RuntimeError("%s: packed image [N=%d G=%d C=%d %dx%d] does not fit N=%d G=%d Cin=%d %dx%d"
                           % (what, img.N, img.G, img.C, img.H, img.W, N, G, Cin, H, W))


def _packed(weight: torch.Tensor, w_rows_tot: int, d: ConvDesc, pl: ConvPlan, mode: int, key) -> torch.Tensor:
    """Packs `weight` for (desc, plan, mode); cached while the SAME tensor object is not modified
    (the entry holds a weak reference: a new tensor that reuses a freed address must not hit)."""
    if not weight.is_leaf:
        # a temporary (the concatenated CRN head weights, crn_model.py): a new tensor every step, so caching its image only
        # grew the cache by two dead entries per step; packed into a scratch buffer of the current stream instead
        buf = torch.empty(int(pl.packed_floats), device=weight.device, dtype=torch.float32)
        check(lib().jaf_conv2d_pack(_s(), ctypes.byref(d), ctypes.byref(pl), mode, _p(weight), w_rows_tot, _p(buf)),
              "jaf_conv2d_pack")
        return buf
    ck = (id(weight), weight.data_ptr(), weight._version, mode, key, pl.MT, pl.CK, pl.precision, pl.nsteps, pl.plane)
    hit = _PACK_CACHE.get(ck)
    if hit is not None and hit.ref() is weight:
        if hit.event is not None:        # image made or refreshed on another stream: order this stream behind it
            # (the raw handle first: building a torch Stream object costs ~12 us, and this runs for every convolution launch)
            sid = _RAW_STREAM(_RAW_DEVICE()) if _RAW_STREAM is not None else torch.cuda.current_stream().cuda_stream
            if sid not in hit.waited:
                cur = torch.cuda.current_stream()
                cur.wait_event(hit.event)
                hit.buf.record_stream(cur)          # ... and keep the block from being reused under this stream's reads
                hit.waited.add(sid)
        return hit.buf
    if len(_PACK_CACHE) >= _PACK_CACHE_MAX:
        # bounded cache with first-in-first-out eviction of the oldest quarter (a stage-4 trainer holds ~700 images;
        # the bound only matters for processes that keep building new models): the dropped images were
        # record_stream-ed on every stream that wrote or read them, so freeing them here is safe
        for ck in list(_PACK_CACHE)[:_PACK_CACHE_MAX // 4]:
            ent = _PACK_CACHE.pop(ck)
            keys = _PACK_KEYS_BY_ID.get(ck[0])
            if keys is not None:
                try:
                    keys.remove(ck)
                except ValueError:
                    pass
                if not keys:
                    _PACK_KEYS_BY_ID.pop(ck[0], None)
    buf = torch.empty(int(pl.packed_floats), device=weight.device, dtype=torch.float32)
    check(lib().jaf_conv2d_pack(_s(), ctypes.byref(d), ctypes.byref(pl), mode, _p(weight), w_rows_tot, _p(buf)),
          "jaf_conv2d_pack")
    ent = _PackEntry(weight, buf, d, pl, mode, w_rows_tot)
    cur = torch.cuda.current_stream()
    ent.event = torch.cuda.Event()          # a frozen network's image may be made on the preparation stream and read
    ent.event.record(cur)                   # on the main one (VGG: target features there, generated frame here)
    ent.waited = {cur.cuda_stream}
    _PACK_CACHE[ck] = ent
    _PACK_KEYS_BY_ID.setdefault(id(weight), []).append(ck)
    return buf


class _PackEntry:
    """One cached packed-weight image and what it takes to make it again in place."""
    __slots__ = ("ref", "buf", "d", "pl", "mode", "rows", "event", "waited")

    def __init__(self, weight, buf, d, pl, mode, rows):
        self.ref, self.buf = weakref.ref(weight), buf
        self.d, self.pl = type(d).from_buffer_copy(d), type(pl).from_buffer_copy(pl)
        self.mode, self.rows = mode, rows
        self.event, self.waited = None, set()


def refresh_packed_weights(params) -> None:
    """Re-packs, into the same buffers, every cached image of `params` right after the optimiser wrote them, on
    a side stream: the ~230 tiny pack launches per step leave the dependent chain (they used to run one by one
    in front of the first convolution that needs each image) and the next use only waits for an event."""
    todo = []
    for t in params:
        for ck in _PACK_KEYS_BY_ID.get(id(t), ()):
            e = _PACK_CACHE.get(ck)
            if e is not None and e.ref() is t:
                todo.append((t, e))
    if not todo:
        return
    main = torch.cuda.current_stream()
    st = aux_stream(2)
    st.wait_stream(main)                      # after the optimiser kernel and every reader of the old images
    L = lib()
    # the bf16-family images of one call (= one module) are re-made by ONE launch over a device-resident argument table,
    # built the first time this exact set of cache entries is seen (243 launches of ~8 us per step before)
    batch = [(t, e) for t, e in todo if e.d.precision != PREC_F32] if _BATCHED_REPACK else []
    table = _repack_table(batch) if len(batch) > 1 else None
    with torch.cuda.stream(st):
        if table is not None:
            check(L.jaf_conv2d_pack_batch(_s(), _p(table[0]), len(batch), table[1]), "jaf_conv2d_pack_batch")
        for t, e in todo:
            if table is None or e.d.precision == PREC_F32:
                check(L.jaf_conv2d_pack(_s(), ctypes.byref(e.d), ctypes.byref(e.pl), e.mode, _p(t), e.rows, _p(e.buf)),
                      "jaf_conv2d_pack")
            if st != main:
                # both blocks were allocated on another stream: the allocator must not hand them out again (cache
                # dropped, model freed) while this stream still reads the weights / writes the image
                e.buf.record_stream(st)
                t.record_stream(st)
        ev = None
        if st != main:
            ev = torch.cuda.Event()
            ev.record(st)
    for _, e in todo:
        e.event, e.waited = ev, (set() if ev is not None else {main.cuda_stream})


_BATCHED_REPACK = True

# Split-K partial sums of the packed weight-gradient kernel (include/jafpro_hip.h, jaf_conv2d_wgrad_packed_ws): one scratch buffer per
# stream that launches weight gradients (launches on a stream run one after the other, and the reduction pass of a launch has read the
# partials before the next launch overwrites them).  Opt-in (set_wgrad_partials): measured neutral on the step
# (profiles/experiments/round4_x4.log); what it buys is a fixed summation order -- bit-reproducible weight gradients -- on the layers
# whose atomic traffic is large enough for the library to take the workspace.
_WGRAD_PARTIALS = False
_WGRAD_WS: dict = {}            # stream handle -> uint8 tensor
_WGRAD_WS_OLD: list = []        # outgrown buffers stay allocated (a launch enqueued earlier may still use them)
_WGRAD_WS_NEED: dict = {}       # (descriptor identity, hidden) -> bytes (0: the layer stays on atomics)


def set_wgrad_partials(on: bool) -> bool:
    """Split-K partial sums instead of fp32 atomics where the library's cost model takes them; returns the previous setting."""
    global _WGRAD_PARTIALS
    prev, _WGRAD_PARTIALS = _WGRAD_PARTIALS, bool(on)
    return prev


def _wgrad_workspace(d, hidden: int, stream_handle: int, device):
    """(pointer or None, bytes) for jaf_conv2d_wgrad_packed_ws."""
    if not _WGRAD_PARTIALS:
        return None, 0
    key = (bytes(d), hidden)          # the descriptor's contents (an id() can be reused by another layer's descriptor: ADVICE r4)
    need = _WGRAD_WS_NEED.get(key)
    if need is None:
        need = int(lib().jaf_conv2d_wgrad_packed_ws_bytes(ctypes.byref(d), hidden))
        need = max(need, 0)
        if len(_WGRAD_WS_NEED) > 8192:
            _WGRAD_WS_NEED.clear()
        _WGRAD_WS_NEED[key] = need
    if need == 0:
        return None, 0
    buf = _WGRAD_WS.get(stream_handle)
    if buf is None or buf.numel() < need or buf.device != device:
        if buf is not None:
            # an outgrown buffer may still be read by a launch enqueued earlier on that stream: the caching allocator keeps the block
            # out of circulation until the stream has passed this point (record_stream), then it is released -- nothing piles up
            for st in ([torch.cuda.current_stream()] + list(_AUX_STREAMS.values())):
                if st.cuda_stream == stream_handle:
                    buf.record_stream(st)
            _WGRAD_WS_OLD[:] = [buf]
        buf = torch.empty(max(need, 64 << 20), dtype=torch.uint8, device=device)
        _WGRAD_WS[stream_handle] = buf
    return _p(buf), buf.numel()
_REPACK_TABLES: dict = {}


def _repack_table(batch):
    """(device table of jaf_conv2d_pack_item entries, largest image's element count) for this list of (weight, cache
    entry) pairs; keyed by the entries' identities and addresses (a re-made entry or a moved weight gets a new table)."""
    key = tuple((id(e), e.buf.data_ptr(), t.data_ptr()) for t, e in batch)
    hit = _REPACK_TABLES.get(key)
    if hit is not None:
        return hit
    L = lib()
    nb = int(L.jaf_conv2d_pack_item_bytes())
    host = (ctypes.c_ubyte * (nb * len(batch)))()
    most = 0
    tot = ctypes.c_int64(0)
    for i, (t, e) in enumerate(batch):
        check(L.jaf_conv2d_pack_item(ctypes.byref(e.d), ctypes.byref(e.pl), e.mode, _p(t), e.rows, _p(e.buf),
                                     ctypes.byref(host, i * nb), ctypes.byref(tot)), "jaf_conv2d_pack_item")
        most = max(most, int(tot.value))
    dev = torch.frombuffer(host, dtype=torch.uint8).clone().to(batch[0][0].device)
    if len(_REPACK_TABLES) > 64:
        _REPACK_TABLES.clear()
    hit = _REPACK_TABLES[key] = (dev, most, [e for _, e in batch])      # the entries stay alive with their table
    return hit


def _out_size(n, k, s, p):
    return (n + 2 * p - k) // s + 1


_AUX_STREAMS = {}
_SERIAL_STREAMS = False


def set_serial_streams(flag: bool) -> bool:
    """Profiling aid: every auxiliary stream becomes the current stream (per-kernel times free of overlap)."""
    global _SERIAL_STREAMS
    prev, _SERIAL_STREAMS = _SERIAL_STREAMS, bool(flag)
    return prev


def aux_stream(which: int = 0) -> "torch.cuda.Stream":
    """One of a few long-lived side HIP streams of the current device (0: clip preparation, 1: weight
    gradients, 2: weight re-packing, 3: the perceptual loss and its frame gradient beside the discriminator phase)."""
    cur = torch.cuda.current_stream()
    if _SERIAL_STREAMS:
        return cur
    k = (cur.device.index, which)
    st = _AUX_STREAMS.get(k)
    if st is None:
        st = torch.cuda.Stream(device=cur.device)
        _AUX_STREAMS[k] = st
    return st


_CHAIN_STREAMS = {}


def chain_stream() -> Optional["torch.cuda.Stream"]:
    """The high-priority HIP stream a trainer runs its dependent chain on (one per device); None while the streams are
    serialised for profiling."""
    if _SERIAL_STREAMS:
        return None
    dev = torch.cuda.current_device()
    st = _CHAIN_STREAMS.get(dev)
    if st is None:
        st = _CHAIN_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=-1)
    return st


_WGRAD_STREAM = None     # see set_wgrad_stream


def set_wgrad_stream(stream):
    """While set, the weight-gradient kernels of convolutions / ConvLSTMs whose parameter owns its .grad buffer
    (step.FlatParams) are enqueued on `stream` instead of the current one: they depend only on the packed dz
    and the saved input, nothing downstream in the backward pass waits for them, so they run beside the data
    gradients of the layers further upstream.  Whoever reads the .grad buffers must call `join_wgrad_stream()`
    first (the trainer does before every Adam / all-reduce and before it returns).  Returns the previous value."""
    global _WGRAD_STREAM
    prev, _WGRAD_STREAM = _WGRAD_STREAM, stream
    if stream is not prev:
        _WGRAD_PENDING[1] = None      # join_wgrad_stream: nobody has joined THIS stream yet
    return prev


_WGRAD_PENDING = [False, None]      # [enqueued on the weight-gradient stream since the last join, the stream that joined last]
_LAZY_JOIN = True

# Operands of a weight-gradient launch (packed input, packed dz) are allocated on the main stream and read on the side stream.
# `record_stream` would hand their blocks back only when the GPU has passed the launch -- with the host up to one and a half steps
# ahead that is ~75 ms later, and every step in flight keeps its own copy of all of them (the caching allocator reserved 2.2 x the
# peak allocation).  Instead the tensors are kept alive on the host until the allocating stream has WAITED for an event recorded
# behind the launch -- at most _WGRAD_KEEP_LAG launches later (by then the side stream has long passed it: the wait costs nothing), or
# at the next join -- and are then freed in stream order like any other tensor: reusable at once, no deferred frees.
_WGRAD_KEEP = collections.deque()       # (event, tensors, allocating stream)
_WGRAD_KEEP_LAG = 12
_KEEP_EVENTS: list = []                 # recycled events


def _wgrad_keep(tensors, ws) -> None:
    if torch.cuda.is_current_stream_capturing():       # (a captured step allocates from the graph's private pool: the old rule)
        for t in tensors:
            if t is not None:
                t.record_stream(ws)
        return
    cur = torch.cuda.current_stream()
    ev = _KEEP_EVENTS.pop() if _KEEP_EVENTS else torch.cuda.Event()
    ev.record(ws)
    _WGRAD_KEEP.append((ev, [t for t in tensors if t is not None], cur))
    while len(_WGRAD_KEEP) > _WGRAD_KEEP_LAG:
        ev0, ts0, st0 = _WGRAD_KEEP.popleft()
        st0.wait_event(ev0)
        ts0.clear()
        _KEEP_EVENTS.append(ev0)


def _wgrad_keep_release(cur) -> None:
    """After `cur` has joined the weight-gradient stream: everything kept for launches whose operands `cur` allocated is free."""
    if not _WGRAD_KEEP:
        return
    rest = [e for e in _WGRAD_KEEP if e[2] != cur]
    for ev0, ts0, st0 in _WGRAD_KEEP:
        if st0 == cur:
            ts0.clear()
            _KEEP_EVENTS.append(ev0)
    _WGRAD_KEEP.clear()
    _WGRAD_KEEP.extend(rest)


def wgrad_stream():
    """The stream set by set_wgrad_stream (None: weight gradients run on the current stream)."""
    return _WGRAD_STREAM


def join_wgrad_stream():
    """The current stream waits for the weight-gradient stream -- only if anything went onto it since this stream's last join: a
    wait on an idle side stream is NOT free here (its marker can sit behind another stream's kernels in a shared hardware queue:
    the trainer's redundant join behind the last optimiser step stalled the main stream 0.85 ms per step)."""
    if _WGRAD_STREAM is None:
        return
    cur = torch.cuda.current_stream()
    if _LAZY_JOIN and not _WGRAD_PENDING[0] and _WGRAD_PENDING[1] == cur:
        return
    cur.wait_stream(_WGRAD_STREAM)
    _WGRAD_PENDING[0] = False
    _WGRAD_PENDING[1] = cur
    _wgrad_keep_release(cur)


_WGRAD_WATCH = [None]       # (ids of the weights still to come, callback): see watch_wgrads


def watch_wgrads(weights, callback=None) -> None:
    """`callback()` runs on the host right after the LAST of `weights` has had its weight gradient enqueued in this backward pass
    (each weight once: the convolution / ConvLSTM backward that owns it reports when its launches are out), whatever order the
    backward pass visits them in.  An event recorded on the weight-gradient stream inside the callback therefore marks 'every
    gradient of this parameter range is complete'.  watch_wgrads(None) clears a watch that has not fired."""
    _WGRAD_WATCH[0] = None if not weights else ({id(w) for w in weights}, callback)
    _WGRAD_FIRED.clear()
    if weights:
        _WGRAD_FIRED_ARMED[0] = {id(w) for w in weights}
    else:
        _WGRAD_FIRED_ARMED[0] = None


_WGRAD_FIRED: set = set()             # ids of the watched weights once the callback has run (until the next watch_wgrads call)
_WGRAD_FIRED_ARMED = [None]


def _wgrad_enqueued(weight) -> None:
    if id(weight) in _WGRAD_FIRED:
        # the callback told its listener "every gradient of this parameter range is complete" (the multi-rank trainer starts the
        # range's all-reduce there): a later weight-gradient launch into the range would race with the message
        raise RuntimeError("a weight gradient was launched after its parameter range had been declared complete")
    w = _WGRAD_WATCH[0]
    if w is not None and id(weight) in w[0]:
        w[0].discard(id(weight))
        if not w[0]:
            _WGRAD_WATCH[0] = None
            if _WGRAD_FIRED_ARMED[0]:
                _WGRAD_FIRED.update(_WGRAD_FIRED_ARMED[0])
            w[1]()


def wgrad_stream_after_current():
    """The weight-gradient stream after it has been made to wait for everything enqueued on the current stream so far (None while
    no such stream is set): work issued on it from here on sees both the weight gradients and the current stream's results up to
    this point -- where dist.BackwardOverlap issues a module's gradient messages, so that the current stream itself never waits."""
    ws = _WGRAD_STREAM
    if ws is None:
        return None
    ws.wait_stream(torch.cuda.current_stream())
    _WGRAD_PENDING[0] = True
    return ws


class LNStats:
    """Side channel between a convolution and the CRN LayerNorm that follows it (crn_model.ConvBlock): on the
    packed bf16 path the conv epilogue accumulates each image's (sum, sum of squares) into `sums`
    ([N][slots][2] fp64) while the outputs are still in registers, and the LayerNorm skips its own
    statistics pass over the tensor.  `filled` stays False when the convolution took another path.
    One object per LayerNorm module: the buffer is zeroed once, `jaf_layernorm_finalize` leaves it clean
    (`dirty` re-zeroes it if an accumulation was never finalised)."""
    __slots__ = ("sums", "slots", "filled", "dirty")
    SLOTS = 8

    def __init__(self):
        self.sums, self.slots, self.filled, self.dirty = None, self.SLOTS, False, False

    def buffer(self, N: int, device) -> torch.Tensor:
        n = N * self.slots * 2
        if self.sums is None or self.sums.numel() != n or self.sums.device != device:
            self.sums = torch.zeros(n, device=device, dtype=torch.float64)
        elif self.dirty:
            self.sums.zero_()
        self.dirty = True
        return self.sums


def _conv_raw(srcs: Sequence[torch.Tensor], specs, weight: torch.Tensor, w_rows_tot: int, mode: int,
              bias: Optional[torch.Tensor], N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil,
              w_cin_tot, w_cin_off, act, slope, out: Optional[torch.Tensor] = None, out_ctot=None, out_coff=0,
              xp: Optional[torch.Tensor] = None, want_xp: bool = False, ln_stats: Optional["LNStats"] = None,
              prepacked: Optional[PackedImage] = None, dst: Optional[PackedDst] = None, skip_f32: bool = False, lazy=None,
              accumulate: bool = False, out2: Optional[torch.Tensor] = None, split: int = 0, dz_fuse=None,
              out_dtype: Optional[torch.dtype] = None):
    """`accumulate`: out += result (packed bf16 path only; `out` must be given): see GradSlot.
    `out_dtype` = torch.bfloat16 (bf16 arithmetic only, see bf16_storage_active): the NCHW result is stored in bf16; a given `out` /
    `out2` says so by its own dtype.
    `dz_fuse`: the launch is a data gradient whose only output is the PRODUCER layer's packed dz in `dst` (jaf_packed_io.dz_mask).
    `out2`, `split`: rows >= split of every group go to out2 [N, G*(Cout-split), OH, OW], the others to `out` taken as
    [N, G*split, OH, OW] (pass out_ctot = G*Cout): jaf_packed_io.out2."""
    skip_f32 = skip_f32 and dst is not None and ln_stats is None
    if accumulate and (out is None or skip_f32):
        raise RuntimeError("conv: accumulate needs an existing fp32 output")
    if out is None:
        out_ctot = G * Cout
        if skip_f32:
            # nothing reads this tensor in fp32: it exists only as the autograd edge (shape / dtype / device), its one
            # element is never written.  Any op that is not packed-aware rejects it (not contiguous).
            # Under bf16 storage the handle is a bf16 tensor: autograd then expects -- and the layers behind it produce -- its
            # gradient in bf16 (a data gradient that is not handed over as a packed dz, a GradSlot buffer).
            out = torch.empty_strided((N, out_ctot, OH, OW), (0, 0, 0, 0), device=srcs[0].device,
                                      dtype=torch.bfloat16 if (bf16_handles_active() and mode == PACK_FWD) else torch.float32)
        else:
            out = torch.empty((N, out_ctot, OH, OW), device=srcs[0].device, dtype=out_dtype or torch.float32)
    out_bf16 = out.dtype == torch.bfloat16 and not skip_f32
    if (out_bf16 or (out2 is not None and out2.dtype == torch.bfloat16)) and _PRECISION != PREC_BF16:
        raise RuntimeError("conv: bf16 output tensors belong to the bf16 arithmetic mode")
    key = (N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, tuple(specs), w_cin_tot, w_cin_off,
           out_ctot, out_coff, act, float(slope))
    d = _make_desc(N, G, Cin, Cout, H, W, OH, OW, KH, KW, stride, pad_t, pad_l, dil, specs, w_cin_tot, w_cin_off,
                   out_ctot, out_coff, act, slope)
    pl = _plan(key, d, 0, 1 if skip_f32 else 0)           # JAF_PLAN_NO_INTERLEAVE for packed-only outputs
    wpk = _packed(weight, w_rows_tot, d, pl, mode, key[:17])
    if _packed_path(d):
        if prepacked is not None:           # the producers already wrote this layer's input image
            _check_image(prepacked, N, G, Cin, H, W, "conv2d")
            xp = prepacked.buf
        elif xp is None:
            xp = pack_input(srcs, d, lazy)
        if dst is not None:
            _check_image(dst.image, dst.image.N, G, 0, OH, OW, "conv2d destination")
            if dst.img_off + N > dst.image.N:
                raise RuntimeError("conv2d destination: images %d..%d outside the packed image (%d)" % (dst.img_off, dst.img_off + N, dst.image.N))
        io = _io_struct(prepacked, dst, skip_f32, accumulate, out2, split, dz_fuse, out_bf16=out_bf16)
        sums = None
        if ln_stats is not None:
            ln_stats.filled = False
            if act == ACT_NONE and G == 1:
                sums = ln_stats.buffer(N, out.device)
        ev = _PROF.begin() if _PROF is not None else None
        check(lib().jaf_conv2d_fwd_packed_io(_s(), ctypes.byref(d), ctypes.byref(pl), _p(xp), _p(wpk), _p(bias),
                                             None if skip_f32 else _p(out),
                                             _p(sums), ln_stats.slots if sums is not None else 1,
                                             ctypes.byref(io) if io is not None else None),
              "jaf_conv2d_fwd_packed_io")
        if sums is not None:
            ln_stats.filled = True
        if ev is not None:
            _PROF.end(_launched(), 2.0 * N * G * Cout * Cin * KH * KW * OH * OW / (dil * dil), ev)
        return (out, xp) if want_xp else out
    if accumulate or out2 is not None:
        raise RuntimeError("conv: accumulate / out2 are features of the packed bf16 path")
    ps = [_p(t) for t in srcs] + [None] * (3 - len(srcs))
    ev = _PROF.begin() if _PROF is not None else None
    check(lib().jaf_conv2d_fwd(_s(), ctypes.byref(d), ctypes.byref(pl), ps[0], ps[1], ps[2], _p(wpk), _p(bias),
                               _p(out)), "jaf_conv2d_fwd")
    if ev is not None:
        _PROF.end(_launched(), 2.0 * N * G * Cout * Cin * KH * KW * OH * OW / (dil * dil), ev)
    return (out, None) if want_xp else out


def _wgrad_packed_ok(m) -> bool:
    return m.KH == m.KW and m.KH in (1, 3, 5, 7) and m.stride in (1, 2)


def _grad_inplace(p: torch.Tensor) -> bool:
    if not p.is_leaf:        # e.g. the concatenated CRN head weights: their gradient goes back through autograd
        return False
    g = p.grad
    return g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.shape == p.shape


class GradSlot:
    """Where the data gradients of a tensor with exactly TWO consumers meet -- both of them convolutions / ConvLSTMs of
    this module (the accumulate network's skip features: the level's ConvLSTM and enc_{i+1}, src/networks.py:1290-1357).
    The consumer whose backward runs first returns its gradient tensor as usual and leaves it here; the second one lets
    its data-gradient kernel add into that buffer (jaf_packed_io.accumulate_f32) and returns no gradient, so the autograd
    engine's separate three-pass add (900 us for the 200 x 200 level at B = 8) disappears.  Attach with `mark_two_consumers`
    ONLY when every consumer of the tensor honours the slot: a third consumer would make the engine replace the buffer."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None

    def take(self, shape):
        """The first gradient, viewed as `shape`, if the second consumer may add into it (bf16 packed path); clears the slot."""
        b, self.buf = self.buf, None
        if b is None or not _packed_path_now() or not b.is_contiguous() or b.numel() != math.prod(shape) or not (
                b.dtype == torch.float32 or (b.dtype == torch.bfloat16 and _PRECISION == PREC_BF16)):
            return None
        return b.view(shape)


def mark_two_consumers(t: torch.Tensor) -> torch.Tensor:
    if t.requires_grad and packed_active():
        t._jaf_gradslot = GradSlot()
    return t


def share_gradslot(src: torch.Tensor, view: torch.Tensor) -> torch.Tensor:
    """`view` (a reshape of `src` handed to the other consumer) meets `src`'s gradients in the same slot."""
    slot = getattr(src, "_jaf_gradslot", None)
    if slot is not None:
        view._jaf_gradslot = slot
        view._jaf_src = src          # the tensor the view was taken of: its producer may take its dz from this consumer
    return view


FUSED_STATS = {"dz": 0, "ln": 0}             # data-gradient launches / LayerNorm backwards that handed their producer a packed dz (tests)
SLOT_STATS = {"first": 0, "added": 0}        # how often a slot received a first gradient / an in-place second one (tests)


def _slot_of(t) -> Optional["GradSlot"]:
    return getattr(t, "_jaf_gradslot", None)


_FUSED_DZ = True


def mark_single_consumer(t: torch.Tensor) -> torch.Tensor:
    """Module code states that `t` (a convolution's output) is read by exactly ONE operation, a convolution: that
    consumer's data-gradient epilogue may then hand the producer its packed dz directly (no fp32 gradient tensor, no
    jaf_conv2d_pack_dz pass)."""
    if t.requires_grad and packed_active():
        t._jaf_single = True
    return t


_DZ_BIAS_SLOTS = 16       # = JAF_DZ_BIAS_SLOTS of include/jafpro_hip.h


def _dz_bias_slots(prod, n: int, device):
    """Slot copies of the producer's bias gradient that a dz-mode launch adds into (None: the producer has no trainable bias)."""
    if prod.has_bias and prod.needs_input_grad[1]:
        return torch.zeros(_DZ_BIAS_SLOTS * n, device=device, dtype=torch.float32)
    return None


def _dz_bias_finish(prod, slots, n: int):
    """Sums the slot copies into the producer's bias gradient: in place when its .grad buffer exists (returns None), else
    into a new tensor that the producer's backward returns."""
    if slots is None:
        return None
    pb = prod.bias_ref
    inplace = _grad_inplace(pb)
    out = pb.grad if inplace else torch.empty(n, device=slots.device, dtype=torch.float32)
    check(lib().jaf_sum_slots(_s(), _p(slots), _DZ_BIAS_SLOTS, n, _p(out), 1 if inplace else 0), "jaf_sum_slots")
    return None if inplace else out


def _fusable_producer(t: torch.Tensor, spec):
    """The _ConvFn node that made `t`, if its activation backward can be taken in its consumer's data-gradient epilogue."""
    fn = t.grad_fn
    if fn is None or type(fn).__name__ != "_ConvFnBackward" or spec[3] == 0:
        return None
    pm = getattr(fn, "meta", None)
    if pm is None or pm.act not in (ACT_LRELU, ACT_RELU) or not _packed_path_now() or getattr(fn, "fwd_key", None) != _fwd_key():
        return None
    if pm.G * pm.Cout != t.shape[1] or spec[0] != pm.Cout:          # the whole output, group for group
        return None
    if fn.needs_input_grad[0] and fn.xp is None:                      # its weight gradient needs an fp32 dz
        return None
    return fn


class _ConvMeta:
    __slots__ = ("G", "stride", "pad", "act", "slope", "shared", "specs", "N", "Cin", "Cout", "H", "W", "OH", "OW",
                 "KH", "KW", "cin_tot", "ln_stats", "prepacked", "dst", "keep_f32", "lazy", "out_dtype")


def _conv_wgrad(ctx, m, weight, srcs, dz, dzp, inplace: bool, stream=None):
    """Weight gradient of one convolution on the CURRENT stream.  A parameter whose .grad buffer already
    exists (step.FlatParams) is accumulated in place by the kernel: no temporary, no memset, no separate
    AccumulateGrad add launch; otherwise the gradient tensor is returned."""
    L = lib()
    dw = weight.grad if inplace else torch.empty_like(weight)
    d = _make_desc(m.N, m.G, m.Cin, m.Cout, m.H, m.W, m.OH, m.OW, m.KH, m.KW, m.stride, m.pad, m.pad, 1,
                   m.specs, m.cin_tot, 0, m.G * m.Cout, 0, ACT_NONE, 0.0)
    ps = [_p(t) for t in srcs] + [None] * (3 - len(srcs))
    # `stream`: launch there without making it torch's current stream (the context manager costs ~25 us per layer); only
    # for launches that allocate nothing (in-place gradient, packed dz given or the fp32-input kernel)
    if stream is not None and (not inplace or (ctx.xp is not None and _packed_path(d) and dzp is None) or _PROF is not None):
        with torch.cuda.stream(stream):
            return _conv_wgrad(ctx, m, weight, srcs, dz, dzp, inplace)
    sh = _s() if stream is None else ctypes.c_void_p(stream.cuda_stream)
    ev = _PROF.begin() if _PROF is not None else None
    if ctx.xp is not None and _packed_path(d):
        if dzp is None:
            dzd = _make_desc(m.N, m.G, m.Cout, 1, m.OH, m.OW, m.OH, m.OW, 1, 1, 1, 0, 0, 1,
                             [(m.Cout, m.G * m.Cout, 0, m.Cout)], 1, 0, m.G, 0, ACT_NONE, 0.0)
            dzp = pack_input([dz], dzd)
        wsp, wsb = _wgrad_workspace(d, 0, sh.value if stream is not None else torch.cuda.current_stream().cuda_stream, dw.device)
        check(L.jaf_conv2d_wgrad_packed_ws_x(sh, ctypes.byref(d), _p(ctx.xp), getattr(ctx, "xp_ng8", 0), 1 if getattr(ctx, "x_split", False) else 0,
                                             _p(dzp), _p(dw), 1 if inplace else 0, 0, wsp, wsb), "jaf_conv2d_wgrad_packed_ws_x")
        wname = _launched() if ev is not None else ""
    else:
        check(L.jaf_conv2d_wgrad(sh, ctypes.byref(d), ps[0], ps[1], ps[2], _p(dz), _p(dw), 1 if inplace else 0),
              "jaf_conv2d_wgrad")
        wname = _launched() if ev is not None else ""
    if ev is not None:
        _PROF.end(wname, 2.0 * m.N * m.G * m.Cout * m.Cin * m.KH * m.KW * m.OH * m.OW, ev)
    return None if inplace else dw


class _ConvFn(Function):
    @staticmethod
    def forward(ctx, weight, bias, meta: _ConvMeta, *srcs):
        m = meta
        use_img = packed_active()
        y, xp = _conv_raw(srcs, m.specs, weight, m.Cout, PACK_FWD, bias, m.N, m.G, m.Cin, m.Cout, m.H, m.W, m.OH, m.OW,
                          m.KH, m.KW, m.stride, m.pad, m.pad, 1, m.cin_tot, 0, m.act, m.slope, want_xp=True,
                          ln_stats=m.ln_stats, prepacked=m.prepacked if use_img else None, dst=m.dst if use_img else None,
                          skip_f32=use_img and not m.keep_f32 and m.dst is not None and m.dst.coff % 8 == 0
                          and m.act in (ACT_LRELU, ACT_RELU), lazy=m.lazy,
                          out_dtype=m.out_dtype if (m.out_dtype is not None and bf16_storage_active()) else None)
        if m.lazy is not None and xp is None:
            raise RuntimeError("conv2d: a lazily resized source reached a convolution outside the packed bf16 path")
        # the activation backward needs only the SIGN of y (ReLU / LeakyReLU), which the consumer's packed bf16 image holds
        # too: read it from there whenever this layer wrote one (2 B per element instead of 4; the only copy when the fp32
        # tensor was skipped)
        ctx.y_img = (m.dst.image, m.dst.coff, m.dst.img_off) if (
            xp is not None and use_img and m.dst is not None and m.dst.coff % 8 == 0 and m.act in (ACT_LRELU, ACT_RELU)
            and _Y_SIGN_FROM_IMAGE) or (y.stride(0) == 0 and y.numel() > 1) else None
        # the packed bf16 input is kept for the weight gradient when the packed wgrad kernel covers the layer
        ctx.xp = xp if (xp is not None and ctx.needs_input_grad[0] and _wgrad_packed_ok(m)) else None
        ctx.xp_ng8 = m.prepacked.ng8 if (use_img and m.prepacked is not None) else 0
        ctx.meta = m
        ctx.mode = _bwd_mode()
        ctx.fwd_key = _fwd_key()
        ctx.x_split = _MIXED            # the saved images are split images read by a bf16 backward pass
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.slots = [_slot_of(t) for t in srcs]
        # backward hand-over in bf16 (jaf_packed_io.dz_mask): sources whose producer is a ReLU / LeakyReLU convolution of the
        # packed path get their dz = dx * act'(x), packed, straight from THIS layer's data-gradient epilogue -- when this layer
        # is the last one to contribute to dx (its only consumer, or the second of a GradSlot pair).  x's sign is read from the
        # packed image this layer consumed.
        ctx.fused = None
        ctx.prods = None
        ctx.mask = None
        if xp is not None and _FUSED_DZ and any(ctx.needs_input_grad[3:]):
            prods = [_fusable_producer(t, spec) for t, spec in zip(srcs, m.specs)]
            if any(p is not None for p in prods):
                ctx.prods = prods
                ctx.single = [bool(getattr(t, "_jaf_single", False)) for t in srcs]
                ctx.mask = (xp, m.prepacked.ng8 if (use_img and m.prepacked is not None) else (m.Cin + 7) // 8)
        ctx.save_for_backward(weight, y if m.act != ACT_NONE else None, *srcs)
        return y

    @staticmethod
    def backward(ctx, dy):
        with _arith(ctx.mode):
            return _ConvFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        m: _ConvMeta = ctx.meta
        weight, y = ctx.saved_tensors[0], ctx.saved_tensors[1]
        srcs = ctx.saved_tensors[2:]
        if getattr(ctx, "fused", None) is None:       # (a handed-over dz comes with a storage-less placeholder for dy)
            dy = _c(dy)
            if dy.dtype != torch.float32 and not (dy.dtype == torch.bfloat16 and _packed_path_now() and _PRECISION == PREC_BF16):
                dy = dy.float()
        L = lib()
        bias = ctx.bias_ref
        want_db = ctx.has_bias and ctx.needs_input_grad[1]
        db = None
        db_done = False
        dzp = None       # packed dz: shared by the data gradients of all sources and the weight gradient
        fused = getattr(ctx, "fused", None)
        if fused is not None:
            # the consumer's data-gradient epilogue already made dz (bf16, packed) and the bias gradient: `dy` is a placeholder
            dzp, db = fused
            ctx.fused = None
            dz = None
            db_done = True
        elif _packed_path_now():
            # one pass: activation backward + bias gradient + packed bf16 dz (+ fp32 dz only if the
            # weight gradient of this layer still runs on the fp32-input kernel); split-bf16: hi and lo planes
            need_f32 = ctx.needs_input_grad[0] and ctx.xp is None
            dz = torch.empty(dy.shape, device=dy.device, dtype=torch.float32) if need_f32 else None
            dyb = 1 if dy.dtype == torch.bfloat16 else 0          # the gradient of a bf16-stored output (bf16_storage_active)
            dbt = None
            if want_db:
                if _grad_inplace(bias):
                    dbt = bias.grad
                else:
                    dbt = db = torch.zeros(m.G * m.Cout, device=dy.device, dtype=torch.float32)
                db_done = True
            ng8 = (m.Cout + 7) // 8
            dzp = torch.empty(m.N * m.G * ng8 * m.OH * m.OW * 16 * (2 if _PRECISION == PREC_BF16X3 else 1), device=dy.device, dtype=torch.uint8)
            yimg = getattr(ctx, "y_img", None)
            with _hbm("conv_pack_dz_kernel", dy.numel() * (dy.element_size() + ((2.0 if yimg else 4.0) if m.act != ACT_NONE else 0.0) + (4.0 if dz is not None else 0.0)) + dzp.numel()):
                per = dzp.numel() // m.N
                for n0, n1 in _n_chunks(m.N, m.G * ng8):
                    if yimg is not None:
                        img, ycoff, yoff = yimg
                        yper = img.per_image
                        check(L.jaf_conv2d_pack_dz_dt2(_s(), _p(dy[n0:n1]), dyb, None, _p(img.buf[(yoff + n0) * yper:(yoff + n1) * yper]),
                                                       img.ng8, ycoff, 1 if img.split else 0, n1 - n0, m.G, m.Cout, m.OH, m.OW, m.act, m.slope,
                                                       _p(dzp[n0 * per:n1 * per]), _p(dz[n0:n1]) if dz is not None else None,
                                                       _p(dbt), _PRECISION), "jaf_conv2d_pack_dz_dt2")
                        continue
                    check(L.jaf_conv2d_pack_dz_dt(_s(), _p(dy[n0:n1]), dyb, _p(y[n0:n1]) if m.act != ACT_NONE else None, None, 0, 0,
                                                  n1 - n0, m.G, m.Cout, m.OH, m.OW, m.act, m.slope, _p(dzp[n0 * per:n1 * per]),
                                                  _p(dz[n0:n1]) if dz is not None else None, _p(dbt), _PRECISION), "jaf_conv2d_pack_dz_dt")
        elif m.act != ACT_NONE:
            dz = torch.empty_like(dy)
            check(L.jaf_act_bwd(_s(), _p(dy), _p(y), _p(dz), dy.numel(), m.act, m.slope), "jaf_act_bwd")
        else:
            dz = dy
        dw = None
        wgrad_done = False
        ws = _WGRAD_STREAM
        if ws is not None and ctx.needs_input_grad[0] and _grad_inplace(weight):
            # weight gradient first, on the side stream: it needs only dz (just made) and the saved input
            ws.wait_stream(torch.cuda.current_stream())
            _WGRAD_PENDING[0] = True
            _conv_wgrad(ctx, m, weight, srcs, dz, dzp, True, stream=ws)
            _wgrad_keep((ctx.xp, dzp, dz) + tuple(srcs), ws)
            wgrad_done = True
        dsrcs: List[Optional[torch.Tensor]] = []
        pad_d = m.KH - 1 - m.pad
        coff = 0
        for i, t in enumerate(srcs):
            c, ctot, _, gs = m.specs[i]
            if ctx.needs_input_grad[3 + i]:
                # transposed convolution: rows = this source's channels, reduction = Cout
                spec = [(m.Cout, m.G * m.Cout, 0, m.Cout)]
                slot = ctx.slots[i] if gs != 0 else None
                first = slot.take((m.N, m.G * c, m.H, m.W)) if slot is not None else None
                prod = ctx.prods[i] if ctx.prods is not None else None
                if prod is not None and coff % 8 == 0 and ((slot is not None and first is not None) or (slot is None and ctx.single[i])):
                    # last contribution to dx: write the producer's packed dz (and bias gradient) instead of an fp32 dx
                    pm = prod.meta
                    dzimg = PackedImage(m.N, m.G, c, m.H, m.W, dy.device)
                    pdb = _dz_bias_slots(prod, m.G * c, dy.device)
                    mbuf, mng8 = ctx.mask
                    g, dzp = _conv_raw([dz] if dz is not None else [dy], spec, weight, m.Cout, PACK_DGRAD, None, m.N, m.G, m.Cout, c,
                                       m.OH, m.OW, m.H, m.W, m.KH, m.KW, 1, pad_d, pad_d, m.stride, m.cin_tot, coff, ACT_NONE, 0.0,
                                       xp=dzp, want_xp=True, out=first, out_ctot=(m.G * c) if first is not None else None,
                                       accumulate=first is not None, dst=dzimg.slot(0, 0, pad_tail=True), skip_f32=first is None,
                                       dz_fuse=(mbuf, mng8, coff, pm.slope if pm.act == ACT_LRELU else 0.0, pdb, getattr(ctx, "x_split", False)))
                    if first is None and g.dtype != t.dtype:
                        g = torch.empty_strided(tuple(g.shape), (0, 0, 0, 0), device=g.device, dtype=t.dtype)     # (placeholder in the edge's type)
                    prod.fused = (dzimg.buf, _dz_bias_finish(prod, pdb, m.G * c))
                    FUSED_STATS["dz"] += 1
                    if first is not None:
                        g = None
                        SLOT_STATS["added"] += 1
                    dsrcs.append(g)      # (single consumer: the storage-less placeholder _conv_raw made)
                    coff += c
                    continue
                # (the gradient takes the source's storage type: bf16 for the bf16-stored tensors of the bf16 mode, see bf16_storage_active)
                g, dzp = _conv_raw([dz] if dz is not None else [dy], spec, weight, m.Cout, PACK_DGRAD, None, m.N, m.G, m.Cout, c, m.OH, m.OW, m.H, m.W,
                                   m.KH, m.KW, 1, pad_d, pad_d, m.stride, m.cin_tot, coff, ACT_NONE, 0.0, xp=dzp,
                                   want_xp=True, out=first, out_ctot=(m.G * c) if first is not None else None,
                                   accumulate=first is not None,
                                   out_dtype=torch.bfloat16 if (t.dtype == torch.bfloat16 and first is None and _packed_path_now() and _PRECISION == PREC_BF16) else None)
                if g.dtype != t.dtype and first is None:
                    g = g.to(t.dtype)
                if gs == 0:      # source shared by all groups: sum the per-group gradients
                    g = g.view(m.N, m.G, c, m.H, m.W).sum(1)
                if first is not None:
                    g = None     # added into the other consumer's gradient (GradSlot)
                    SLOT_STATS["added"] += 1
                elif slot is not None:
                    slot.buf = g
                    SLOT_STATS["first"] += 1
                dsrcs.append(g)
            else:
                dsrcs.append(None)
            coff += c
        if ctx.needs_input_grad[0] and not wgrad_done:
            inplace = _grad_inplace(weight)
            dw = _conv_wgrad(ctx, m, weight, srcs, dz, dzp, inplace)
        if want_db and not db_done:
            inplace = bias is not None and _grad_inplace(bias)
            db = bias.grad if inplace else torch.empty(m.G * m.Cout, device=dy.device, dtype=torch.float32)
            check(L.jaf_channel_sum(_s(), _p(dz), m.N, m.G * m.Cout, 0, m.G * m.Cout, m.OH * m.OW, _p(db),
                                    1 if inplace else 0), "jaf_channel_sum")
            if inplace:
                db = None
        _wgrad_enqueued(weight)
        return (dw, db, None) + tuple(dsrcs)


def conv2d(srcs, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 1, pad: int = 0,
           act: int = ACT_NONE, slope: float = 0.0, groups: int = 1, shared: Optional[Sequence[bool]] = None,
           ln_stats: Optional[LNStats] = None, prepacked: Optional[PackedImage] = None, dst: Optional[PackedDst] = None,
           keep_f32: bool = True, out_dtype: Optional[torch.dtype] = None):
    """Grouped convolution over the channel concatenation of `srcs` with fused bias + activation.
    `ln_stats`: see LNStats (filled only on the packed bf16 path with act NONE and groups 1).
    `prepacked`: the packed bf16 image of exactly this concatenation, already written by the producers of `srcs`
    (then `srcs` are only the autograd edges); `dst`: the consumer's image slot the outputs are also written to.
    keep_f32=False with a `dst` (ReLU / LeakyReLU layers): every consumer reads the packed image, so the fp32 result
    is not written at all and the returned tensor is a storage-less autograd handle.  All three are honoured on the packed
    bf16 path only (see PackedImage).  out_dtype=torch.bfloat16: the caller's consumers of the result all read bf16 (today: the
    CRN LayerNorm) -- honoured while bf16_storage_active(), fp32 otherwise.

    srcs[i]: [N, groups*c_i, H, W] (or [N, c_i, H, W] when shared[i]: every group reads the same
    channels).  weight: [groups*Cout, sum(c_i), KH, KW] or [groups, Cout, sum(c_i), KH, KW].
    """
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    if not (prepacked is not None and packed_active()):       # with their packed image given, the sources are only autograd edges
        srcs = [t if getattr(t, "_jaf_lazy", None) is not None and _packed_path_now() else _chk(t, "conv2d source") for t in srcs]
    if out_dtype not in (None, torch.float32, torch.bfloat16):
        raise ValueError("conv2d: out_dtype must be torch.float32 or torch.bfloat16")
    _chk(weight, "conv2d weight")
    if bias is not None:
        _chk(bias, "conv2d bias")
    G = groups
    if shared is None:
        shared = [False] * len(srcs)
    m = _ConvMeta()
    KH, KW = int(weight.shape[-2]), int(weight.shape[-1])
    cin_tot = int(weight.shape[-3])
    Cout = int(weight.numel() // (cin_tot * KH * KW)) // G
    N, _, H, W = srcs[0].shape
    specs = []
    for t, sh in zip(srcs, shared):
        if t.shape[0] != N or t.shape[2] != H or t.shape[3] != W:
            raise RuntimeError("conv2d sources disagree in shape")
        ct = int(t.shape[1])
        if sh:
            specs.append((ct, ct, 0, 0))
        else:
            if ct % G:
                raise RuntimeError("source channels not divisible by groups")
            specs.append((ct // G, ct, 0, ct // G))
    Cin = sum(s[0] for s in specs)
    if Cin != cin_tot:
        raise RuntimeError("conv2d: weight expects %d input channels, sources provide %d" % (cin_tot, Cin))
    m.G, m.stride, m.pad, m.act, m.slope = G, stride, pad, act, float(slope)
    m.specs, m.N, m.Cin, m.Cout, m.H, m.W = specs, int(N), Cin, Cout, int(H), int(W)
    m.OH, m.OW, m.KH, m.KW, m.cin_tot = _out_size(H, KH, stride, pad), _out_size(W, KW, stride, pad), KH, KW, cin_tot
    m.ln_stats = ln_stats
    m.prepacked, m.dst, m.keep_f32 = prepacked, dst, keep_f32
    m.out_dtype = out_dtype if (out_dtype == torch.bfloat16 and act == ACT_NONE) else None      # (the activation backward reads y in fp32)
    m.lazy = [getattr(t, "_jaf_lazy", None) for t in srcs]
    if not any(l is not None for l in m.lazy):
        m.lazy = None
    elif not _wgrad_packed_ok(m):
        raise RuntimeError("conv2d: lazily resized sources need a layer whose weight gradient runs on the packed kernel")
    return _ConvFn.apply(weight, bias, m, *srcs)


def conv_transpose2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride: int = 2, pad: int = 1,
                     act: int = ACT_NONE, slope: float = 0.0) -> torch.Tensor:
    """nn.ConvTranspose2d (weight [Cin, Cout, KH, KW], output_padding 0, groups 1) with fused bias + activation, FORWARD ONLY
    (the FlowNetSD evaluation network, src/flownet2_pytorch/networks/submodules.py:34-38).  A transposed convolution is the
    data gradient of the convolution whose weight tensor it shares, so this is the data-gradient launch of `conv2d`
    (zero-dilated input, flipped / transposed packed weights) with an epilogue."""
    _chk(x, "conv_transpose2d input"); _chk(weight, "conv_transpose2d weight")
    if bias is not None:
        _chk(bias, "conv_transpose2d bias")
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        raise RuntimeError("conv_transpose2d is forward-only: call it under torch.no_grad()")
    N, Cin, H, W = (int(v) for v in x.shape)
    if weight.dim() != 4 or weight.shape[0] != Cin:
        raise RuntimeError("conv_transpose2d: weight must be [Cin, Cout, KH, KW]")
    Cout, KH, KW = int(weight.shape[1]), int(weight.shape[2]), int(weight.shape[3])
    if KH != KW or pad > KH - 1:
        raise RuntimeError("conv_transpose2d: square kernels with pad <= k - 1 only")
    OH, OW = (H - 1) * stride - 2 * pad + KH, (W - 1) * stride - 2 * pad + KW
    pad_d = KH - 1 - pad
    return _conv_raw([x], [(Cin, Cin, 0, Cin)], weight, Cin, PACK_DGRAD, bias, N, 1, Cin, Cout, H, W, OH, OW, KH, KW, 1, pad_d, pad_d,
                     stride, Cout, 0, act, float(slope))


def conv2d_direct(srcs, weight, bias=None, stride=1, pad=0, act=ACT_NONE, slope=0.0, groups=1):
    """One-thread-per-output cross-check kernel (tests only)."""
    if isinstance(srcs, torch.Tensor):
        srcs = [srcs]
    G = groups
    KH, KW = int(weight.shape[-2]), int(weight.shape[-1])
    cin_tot = int(weight.shape[-3])
    Cout = int(weight.numel() // (cin_tot * KH * KW)) // G
    N, _, H, W = srcs[0].shape
    specs = [(int(t.shape[1]) // G, int(t.shape[1]), 0, int(t.shape[1]) // G) for t in srcs]
    OH, OW = _out_size(H, KH, stride, pad), _out_size(W, KW, stride, pad)
    d = _make_desc(int(N), G, cin_tot, Cout, int(H), int(W), OH, OW, KH, KW, stride, pad, pad, 1, specs, cin_tot, 0,
                   G * Cout, 0, act, slope)
    out = torch.empty((N, G * Cout, OH, OW), device=srcs[0].device, dtype=torch.float32)
    ps = [_p(t) for t in srcs] + [None] * (3 - len(srcs))
    check(lib().jaf_conv2d_fwd_direct(_s(), ctypes.byref(d), ps[0], ps[1], ps[2], _p(weight), _p(bias), _p(out)),
          "jaf_conv2d_fwd_direct")
    return out


# --------------------------------------------------------------------------------------------
# ConvLSTM (whole sequence, one autograd node; BPTT in backward)
# --------------------------------------------------------------------------------------------
# widest [x, h] data gradient taken in one launch (see _ConvLSTMFn._backward); measured: a gain up to 48 rows (levels with 12 and 24 hidden channels), none above
_LSTM_FUSED_DGRAD_MAX_ROWS = 48


class _StickyBuffers:
    """The ConvLSTM's largest per-call tensors -- the saved gates (2.8 GB at the 200 x 200 level of a B = 8 step) and the cell states --
    kept from one call of a layer to the next instead of going back to the caching allocator.  Both are written and read on the
    stream the layer runs on only (cell kernel, gate backward), so the next call's kernels are ordered behind the last reader by
    the stream itself.  Why: with two steps in flight the allocator now and then finds no free block for the 2.8 GB request and
    answers with a fresh hipMalloc of that size -- 80-90 ms of HOST time, a 75-120 ms step every few steps
    (profiles/experiments/r5_spikes.py).  A buffer is handed out again only when its previous user is done with it: the node that
    took it has run its backward pass or has been destroyed; otherwise (two forward passes before a backward) the caller falls back
    to an ordinary allocation."""

    def __init__(self):
        self.slots = {}

    def take(self, owner, key, shape, dtype, device, anchor=None):
        """`anchor`: the object the key's identity comes from (the layer's weight): entries whose anchor has died are dropped."""
        for k in [k for k, e in self.slots.items() if e[3] is not None and e[3]() is None]:
            del self.slots[k]
        ent = self.slots.get(key)
        if ent is not None and (ent[1] is None or ent[1]() is None) and ent[0].shape == tuple(shape) and ent[0].dtype == dtype \
                and ent[0].device == device and ent[2] == torch.cuda.current_stream(device).cuda_stream:
            self.slots[key] = (ent[0], weakref.ref(owner), ent[2], ent[3])
            return ent[0]
        if ent is not None and ent[1] is not None and ent[1]() is not None:
            return torch.empty(shape, device=device, dtype=dtype)           # still owned by a node whose backward has not run
        t = torch.empty(shape, device=device, dtype=dtype)
        if len(self.slots) > 64:
            self.slots.clear()
        self.slots[key] = (t, weakref.ref(owner), torch.cuda.current_stream(device).cuda_stream,
                           weakref.ref(anchor) if anchor is not None else None)
        return t

    def release(self, owner):
        for k, ent in list(self.slots.items()):
            if ent[1] is not None and ent[1]() is owner:
                self.slots[k] = (ent[0], None, ent[2], ent[3])


_STICKY = _StickyBuffers()
_STICKY_MIN_BYTES = 256 << 20


class _LstmOwner:
    """What a ConvLSTM node holds on to while it still needs its sticky buffers (weakly referenced by _StickyBuffers)."""
    __slots__ = ("__weakref__",)


class _ConvLSTMFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, G: int, need_all: bool, h0, c0, seq_image=None, final_dst=None, want_c: bool = True):
        # x: [T, N, G*C, H, W]; weight: [G*4C, 2C, 3, 3]; bias [G*4C]; h0/c0: [N, G*C, H, W] or None (zero state)
        # seq_image (packed bf16 path only): PackedImage of T*N images x 2C channels whose x halves the producer of `x`
        # already wrote; step t reads image block t and its epilogue writes h_t into block t+1 (final_dst: where h_T goes)
        T, N, GC, H, W = x.shape
        C = GC // G
        L = lib()
        keep = any(ctx.needs_input_grad[:3]) or ctx.needs_input_grad[5] or ctx.needs_input_grad[6]
        ctx.set_materialize_grads(False)
        # on the packed bf16 path the saved gates are bf16 (they are the dominant traffic of the cell epilogue
        # and of the gate backward, which reads them exactly once)
        # ("mixed": the forward pass is split-bf16, but the gates are saved for a bf16 backward pass -- bf16 gates there too)
        g16 = _USE_PACKED and (_PRECISION == PREC_BF16 or _MIXED) and C % 4 == 0
        # bf16 storage: the cell state (written once, read by the next step and twice by the gate backward) in bf16 too -- whenever
        # it stays inside this node (zero initial state, c_T not handed out)
        st16 = g16 and bf16_storage_active() and h0 is None and not want_c
        use_img = seq_image is not None and packed_active() and h0 is None
        # with the images, h_t for t < T-1 exists only as bf16 inside the next step's image (nothing reads it in fp32:
        # the fused gate backward needs c_t and the gates, the weight gradient the packed images)
        skip_h = use_img and g16 and not need_all
        owner = ctx.sticky = _LstmOwner()

        def big(tag, shape, dtype):
            n = math.prod(shape) * (2 if dtype == torch.bfloat16 else 4)
            if n < _STICKY_MIN_BYTES or not weight.is_leaf or torch.cuda.is_current_stream_capturing():
                return torch.empty(shape, device=x.device, dtype=dtype)
            return _STICKY.take(owner, (id(weight), tag), tuple(shape), dtype, x.device, anchor=weight)
        # (only h_T is ever written when the intermediate states are skipped: one step's worth of memory instead of T)
        hs = torch.empty((1 if skip_h else T, N, GC, H, W), device=x.device, dtype=torch.float32)
        cs = big("c", (T, N, GC, H, W), torch.bfloat16 if st16 else torch.float32)
        gates = big("gates", (T, N, 4 * GC, H, W), torch.bfloat16 if g16 else torch.float32) if keep else None
        xps = []         # packed (x_t, h_{t-1}) images: reused by the weight gradient
        if use_img:
            _check_image(seq_image, T * N, G, 2 * C, H, W, "convlstm")
        for t in range(T):
            first = t == 0 and h0 is None
            hprev = h0 if t == 0 else (None if skip_h else hs[t - 1])
            ht = hs[0] if skip_h else hs[t]              # (skip_h: steps before the last write no fp32 h at all)
            cprev = c0 if t == 0 else cs[t - 1]
            specs = [(C, GC, 0, C)] if first else [(C, GC, 0, C), (C, GC, 0, C)]
            Cin = C if first else 2 * C
            key = ("lstm", N, G, Cin, C, H, W)
            d = _make_desc(N, G, Cin, 4 * C, H, W, H, W, 3, 3, 1, 1, 1, 1, specs, 2 * C, 0, 4 * GC, 0, ACT_NONE, 0.0)
            pl = _plan(key, d, 1)
            wpk = _packed(weight, 4 * C, d, pl, PACK_LSTM, key)
            ev = _PROF.begin() if _PROF is not None else None
            if _packed_path(d):
                io = None
                if use_img:
                    xp = seq_image.images(t * N, N).buf
                    hdst = seq_image.slot(C, (t + 1) * N) if t + 1 < T else final_dst
                    if hdst is not None:
                        # n + img_off indexes the destination image from ITS base; the input block starts at t*N
                        io = _io_struct(seq_image, hdst, skip_f32=(skip_h and t + 1 < T), state_bf16=st16)
                    else:
                        io = _io_struct(seq_image, None, state_bf16=st16)
                else:
                    xp = pack_input([x[t]] if first else [x[t], hprev], d)
                    if final_dst is not None and t + 1 == T and packed_active():
                        io = _io_struct(None, final_dst, state_bf16=st16)
                    elif st16:
                        io = _io_struct(None, None, state_bf16=True)
                if keep:
                    xps.append(xp)
                check(L.jaf_convlstm_cell_fwd_packed_io(_s(), ctypes.byref(d), ctypes.byref(pl), _p(xp), _p(wpk), _p(bias),
                                                        None if first else _p(cprev), _p(ht), _p(cs[t]),
                                                        _p(gates[t]) if keep else None, 1 if g16 else 0,
                                                        ctypes.byref(io) if io is not None else None),
                      "jaf_convlstm_cell_fwd_packed_io")
                if ev is not None:
                    _PROF.end(_launched(), 2.0 * N * G * 4 * C * Cin * 9 * H * W, ev)
                continue
            check(L.jaf_convlstm_cell_fwd(_s(), ctypes.byref(d), ctypes.byref(pl), _p(x[t]),
                                          None if first else _p(hprev), _p(wpk), _p(bias),
                                          None if first else _p(cprev), _p(hs[t]), _p(cs[t]),
                                          _p(gates[t]) if keep else None), "jaf_convlstm_cell_fwd")
            if ev is not None:
                _PROF.end(_launched(), 2.0 * N * G * 4 * C * Cin * 9 * H * W, ev)
        ctx.G = G
        ctx.need_all = need_all
        ctx.bias_ref = bias
        ctx.xps = xps if (keep and len(xps) == T) else None
        ctx.has_state = h0 is not None
        ctx.xp_ng8 = seq_image.ng8 if use_img else 0
        ctx.h_skipped = skip_h
        ctx.st16 = st16
        ctx.mode = _bwd_mode()                      # backward uses the arithmetic the forward ran in ("mixed": bf16)
        ctx.fwd_key = _fwd_key()
        ctx.x_split = _MIXED
        if keep:
            ctx.save_for_backward(x, weight, hs, cs, gates, h0, c0)
        ctx.slot = _slot_of(x)
        # the layer that made x (enc_i) may get its packed dz from this node's d x launches (see _ConvFn.forward)
        src = getattr(x, "_jaf_src", None)
        ctx.prod = None
        if use_img and keep and src is not None and _FUSED_DZ and ctx.slot is not None and src.dim() == 4 and src.shape[0] == T * N:
            ctx.prod = _fusable_producer(src, (C, GC, 0, C))
        c_last = cs[T - 1].clone() if want_c else None          # (a copy: cs is saved for backward)
        if not keep:
            _STICKY.release(owner)
        if need_all:
            return hs, c_last
        return hs[-1], c_last

    @staticmethod
    def backward(ctx, dh_out, dc_last=None):
        with _arith(ctx.mode):
            return _ConvLSTMFn._backward(ctx, dh_out, dc_last)

    @staticmethod
    def _backward(ctx, dh_out, dc_last):
        x, weight, hs, cs, gates, h0, c0 = ctx.saved_tensors
        G = ctx.G
        T, N, GC, H, W = x.shape
        C = GC // G
        L = lib()
        if dh_out is None:          # only c_T was used downstream
            dh_out = torch.zeros_like(hs) if ctx.need_all else torch.zeros_like(hs[0])
        dh_out = _c(dh_out)
        # (dx takes x's storage type: bf16 when x is the bf16 handle of an encoder layer under bf16 storage)
        dx = torch.empty(x.shape, device=x.device, dtype=x.dtype if (x.dtype == torch.bfloat16 and ctx.xps is not None) else torch.float32) \
            if ctx.needs_input_grad[0] else None
        # x has a second consumer (enc_{i+1}) whose data gradient may already exist: add into it (GradSlot)
        slot = getattr(ctx, "slot", None)
        dx_first = None
        if dx is not None and slot is not None and ctx.xps is not None and _packed_path_now():
            dx_first = slot.take(tuple(x.shape))
            if dx_first is not None:
                dx = dx_first
        bias = ctx.bias_ref
        w_inplace, b_inplace = _grad_inplace(weight), _grad_inplace(bias)
        dw = weight.grad if w_inplace else torch.empty_like(weight)
        # fused path: gate backward writes the packed bf16 gate gradients + the bias sums directly
        # (split-bf16 too: the saved gates are fp32 there, the packed gate gradients carry hi + lo planes)
        fused = ctx.xps is not None and _packed_path_now() and C % 4 == 0
        if (gates.dtype == torch.bfloat16 or ctx.h_skipped) and not fused:
            raise RuntimeError("convlstm: precision changed between forward and backward")
        st16 = bool(getattr(ctx, "st16", False))       # bf16 storage of c (forward) -> dc and dh of the time loop in bf16 too
        sdt = torch.bfloat16 if st16 else torch.float32
        if b_inplace:
            db = bias.grad
        elif fused:
            db = torch.zeros(4 * GC, device=x.device, dtype=torch.float32)
        else:
            db = torch.empty(4 * GC, device=x.device, dtype=torch.float32)
        dc = _c(dc_last) if dc_last is not None else None
        dh = None
        ng8 = (4 * C + 7) // 8
        # last contribution to dx (the other consumer's gradient is already in dx_first): hand the producer of x its packed dz
        prod = getattr(ctx, "prod", None) if (dx_first is not None and fused) else None
        dzimg = pdb = None
        if prod is not None:
            dzimg = PackedImage(T * N, G, C, H, W, x.device)
            pdb = _dz_bias_slots(prod, GC, x.device)
            pslope = prod.meta.slope if prod.meta.act == ACT_LRELU else 0.0
        for t in range(T - 1, -1, -1):
            first = t == 0 and not ctx.has_state
            hprev = h0 if t == 0 else (hs[t - 1] if hs.shape[0] == T else None)     # (None: h_{t-1} was never written in fp32, and the fused path below never reads it)
            cprev = c0 if t == 0 else cs[t - 1]
            if ctx.need_all:
                dht = dh_out[t] if dh is None else dh_out[t] + dh
            else:
                dht = dh_out if t == T - 1 else dh
            dht = _c(dht)
            dc_prev = torch.empty((N, GC, H, W), device=x.device, dtype=sdt)
            gt = gates[t]
            specs = [(C, GC, 0, C)] if first else [(C, GC, 0, C), (C, GC, 0, C)]
            Cin = C if first else 2 * C
            d = _make_desc(N, G, Cin, 4 * C, H, W, H, W, 3, 3, 1, 1, 1, 1, specs, 2 * C, 0, 4 * GC, 0, ACT_NONE, 0.0)
            acc = 0 if t == T - 1 else 1
            gspec = [(4 * C, 4 * GC, 0, 4 * C)]
            gtp = None       # packed gate gradients: shared by the weight gradient and the two data gradients
            if fused:
                gtp = torch.empty(N * G * ng8 * H * W * 16 * (2 if _PRECISION == PREC_BF16X3 else 1), device=x.device, dtype=torch.uint8)
                if dc is not None and dc.dtype != sdt:
                    dc = dc.to(sdt)
                with _hbm("lstm_gates_bwd_pack_kernel", N * G * C * H * W * (dht.element_size() + (3.0 if first else 4.0) * dc_prev.element_size()
                                                                             + 4.0 * gt.element_size()) + gtp.numel()):
                    check(L.jaf_convlstm_gates_bwd_packed_dt(_s(), N, G, C, H * W, _p(dht), 1 if dht.dtype == torch.bfloat16 else 0, _p(dc), _p(gt),
                                                             1 if gt.dtype == torch.bfloat16 else 0, None if first else _p(cprev), _p(cs[t]),
                                                             _p(dc_prev), 1 if st16 else 0, _p(gtp), _p(db), _PRECISION), "jaf_convlstm_gates_bwd_packed_dt")
                wst = _WGRAD_STREAM if w_inplace else None      # see set_wgrad_stream
                if wst is not None:
                    wst.wait_stream(torch.cuda.current_stream())
                    _WGRAD_PENDING[0] = True
                with torch.cuda.stream(wst if wst is not None else torch.cuda.current_stream()):
                    ev = _PROF.begin() if _PROF is not None else None
                    # (the packed gate gradients are channel-major, 4 c + gate: the kernel permutes the rows of dW)
                    wsp, wsb = _wgrad_workspace(d, C, torch.cuda.current_stream().cuda_stream, dw.device)
                    check(L.jaf_conv2d_wgrad_packed_ws_x(_s(), ctypes.byref(d), _p(ctx.xps[t]), ctx.xp_ng8, 1 if getattr(ctx, "x_split", False) else 0,
                                                         _p(gtp), _p(dw), 1 if w_inplace else acc, C, wsp, wsb), "jaf_conv2d_wgrad_packed_ws_x")
                    if ev is not None:
                        _PROF.end(_launched(), 2.0 * N * G * 4 * C * Cin * 9 * H * W, ev)
                if wst is not None:
                    _wgrad_keep((gtp, ctx.xps[t]), wst)
            else:
                # gt is overwritten with the pre-activation gate gradients
                check(L.jaf_convlstm_gates_bwd(_s(), N, G, C, H * W, _p(dht), _p(dc), _p(gt),
                                               None if first else _p(cprev), _p(cs[t]), _p(dc_prev)),
                      "jaf_convlstm_gates_bwd")
                ev = _PROF.begin() if _PROF is not None else None
                if ctx.xps is not None and _packed_path(d):
                    gd = _make_desc(N, G, 4 * C, 1, H, W, H, W, 1, 1, 1, 0, 0, 1, gspec, 1, 0, G, 0, ACT_NONE, 0.0)
                    gtp = pack_input([gt], gd)
                    check(L.jaf_conv2d_wgrad_packed_ex(_s(), ctypes.byref(d), _p(ctx.xps[t]), ctx.xp_ng8, _p(gtp), _p(dw),
                                                       1 if w_inplace else acc), "jaf_conv2d_wgrad_packed_ex")
                    wname = _launched() if ev is not None else ""
                else:
                    check(L.jaf_conv2d_wgrad(_s(), ctypes.byref(d), _p(x[t]), None if first else _p(hprev), None,
                                             _p(gt), _p(dw), 1 if w_inplace else acc), "jaf_conv2d_wgrad")
                    wname = _launched() if ev is not None else ""
                if ev is not None:
                    _PROF.end(wname, 2.0 * N * G * 4 * C * Cin * 9 * H * W, ev)
                check(L.jaf_channel_sum(_s(), _p(gt), N, 4 * GC, 0, 4 * GC, H * W, _p(db), 1 if b_inplace else acc),
                      "jaf_channel_sum")
            dmode = PACK_DGRAD_LSTM if fused else PACK_DGRAD       # (fused: gtp is channel-major, see jaf_convlstm_gates_bwd_packed)
            if dx is not None and not first and fused and 2 * C <= _LSTM_FUSED_DGRAD_MAX_ROWS:
                # d[x_t, h_{t-1}] in one launch: 2C rows per group, the x rows into dx[t], the h rows into dh -- the packed gate
                # gradients (4C channels) are read once instead of twice
                dh = torch.empty((N, GC, H, W), device=x.device, dtype=sdt)
                fz = None if prod is None else dict(dst=dzimg.images(t * N, N).slot(0, 0, pad_tail=True),
                                                    dz_fuse=(ctx.xps[t], ctx.xp_ng8, 0, pslope, pdb, getattr(ctx, "x_split", False)))
                _, gtp = _conv_raw([gt], gspec, weight, 4 * C, dmode, None, N, G, 4 * C, 2 * C, H, W, H, W, 3, 3, 1, 1, 1,
                                   1, 2 * C, 0, ACT_NONE, 0.0, out=dx[t], out_ctot=2 * GC, out_coff=0, xp=gtp, want_xp=True,
                                   accumulate=dx_first is not None, out2=dh, split=C, **(fz or {}))
            else:
                if dx is not None:
                    fz = None if prod is None else dict(dst=dzimg.images(t * N, N).slot(0, 0, pad_tail=True),
                                                        dz_fuse=(ctx.xps[t], ctx.xp_ng8, 0, pslope, pdb, getattr(ctx, "x_split", False)))
                    _, gtp = _conv_raw([gt], gspec, weight, 4 * C, dmode, None, N, G, 4 * C, C, H, W, H, W, 3, 3, 1, 1, 1,
                                       1, 2 * C, 0, ACT_NONE, 0.0, out=dx[t], out_ctot=GC, out_coff=0, xp=gtp, want_xp=True,
                                       accumulate=dx_first is not None, **(fz or {}))
                if not first:
                    dh = _conv_raw([gt], gspec, weight, 4 * C, dmode, None, N, G, 4 * C, C, H, W, H, W, 3, 3, 1, 1, 1,
                                   1, 2 * C, C, ACT_NONE, 0.0, xp=gtp, out_dtype=sdt)
            dc = dc_prev
        dh0 = dh if (ctx.has_state and ctx.needs_input_grad[5]) else None
        dc0 = dc if (ctx.has_state and ctx.needs_input_grad[6]) else None
        if prod is not None:
            prod.fused = (dzimg.buf, _dz_bias_finish(prod, pdb, GC))
            FUSED_STATS["dz"] += 1
        if dx_first is not None:
            dx = None            # added into the other consumer's gradient
            SLOT_STATS["added"] += 1
        elif dx is not None and slot is not None:
            slot.buf = dx
            SLOT_STATS["first"] += 1
        _wgrad_enqueued(weight)
        owner = getattr(ctx, "sticky", None)
        if owner is not None:
            _STICKY.release(owner)          # (a second backward through this node would read buffers the next call may have overwritten:
            ctx.sticky = None               #  autograd frees the saved tensors after the first one anyway)
        return dx, (None if w_inplace else dw), (None if b_inplace else db), None, None, dh0, dc0, None, None, None


def convlstm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, groups: int = 1, return_all: bool = False,
             state: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, seq_image: Optional[PackedImage] = None,
             final_dst: Optional[PackedDst] = None, return_state: bool = True):
    """x: [T, N, G*C, H, W] -> (h_T [N, G*C, H, W] or all h_t, c_T).  `state` = (h0, c0), default the zero
    state of src/convLSTM.py:58-63,119-120; gate order i,f,o,g (:46).  Differentiable w.r.t. x, the parameters,
    the initial state, and through both h and c_T (T = 1 with a state is ConvLSTMCell.forward, :41-56).
    `return_state=False`: c_T is not wanted (-> None), which saves its copy."""
    if not (seq_image is not None and packed_active() and state is None):     # x is then only the autograd edge
        _chk(x, "convlstm x")
    _chk(weight, "convlstm weight"); _chk(bias, "convlstm bias")
    h0 = c0 = None
    if state is not None:
        h0, c0 = _chk(state[0], "convlstm h0"), _chk(state[1], "convlstm c0")
        if h0.shape != x.shape[1:] or c0.shape != x.shape[1:]:
            raise RuntimeError("convlstm: state shape %s / %s does not match the input %s" % (tuple(h0.shape), tuple(c0.shape), tuple(x.shape[1:])))
    return _ConvLSTMFn.apply(x, weight, bias, groups, return_all, h0, c0, seq_image, final_dst, return_state)


# --------------------------------------------------------------------------------------------
# normalisation
# --------------------------------------------------------------------------------------------
def _ln_producer(x: torch.Tensor):
    """The _ConvFn node that made `x`, if a LayerNorm that is its only reader may hand it dz in packed bf16
    (jaf_layernorm_lrelu_bwd_packed): a convolution without activation on the packed bf16 path, one group, whose weight
    gradient (if any) runs on the packed kernel."""
    fn = x.grad_fn
    if not _FUSED_DZ or fn is None or type(fn).__name__ != "_ConvFnBackward":
        return None
    pm = getattr(fn, "meta", None)
    if pm is None or pm.act != ACT_NONE or pm.G != 1 or pm.Cout != x.shape[1] or getattr(fn, "fwd_key", None) != _fwd_key():
        return None
    if getattr(fn, "y_img", None) is not None or (fn.needs_input_grad[0] and fn.xp is None):
        return None
    return fn


class _LayerNormLReLUFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, slope: float, pre: Optional[LNStats], dst=None, keep_f32=True, sole=False):
        N, C, H, W = x.shape
        L = lib()
        # backward hand-over: x is a convolution's output and this LayerNorm its only reader -> dx goes back as that layer's
        # packed dz (+ bias gradient), no fp32 dx, no jaf_conv2d_pack_dz pass
        ctx.prod = _ln_producer(x) if (sole and ctx.needs_input_grad[0] and _packed_path_now()) else None
        stats = torch.empty(2 * N, device=x.device, dtype=torch.float32)
        xb = 1 if x.dtype == torch.bfloat16 else 0           # bf16 storage of the pre-LayerNorm convolution output
        if pre is not None and pre.filled:       # sums came out of the producing convolution's epilogue (from the unrounded values)
            check(L.jaf_layernorm_finalize(_s(), _p(pre.sums), N, pre.slots, C * H * W, eps, _p(stats)), "jaf_layernorm_finalize")
            pre.filled, pre.dirty = False, False     # consumed, and the kernel left the sums at zero
        else:
            if xb:
                raise RuntimeError("layernorm: a bf16-stored input must come with the statistics of its producing convolution")
            ws = torch.empty(2 * N, device=x.device, dtype=torch.float64)
            check(L.jaf_layernorm_stats(_s(), _p(x), N, C * H * W, eps, _p(ws), _p(stats)), "jaf_layernorm_stats")
        if dst is not None and packed_active():
            _check_image(dst.image, N, 1, 0, H, W, "layernorm destination")
            if dst.img_off or dst.coff % 8 or dst.coff + C > dst.image.ng8 * 8:
                raise RuntimeError("layernorm destination: slot does not fit")
            # keep_f32=False: only the convolution behind `dst` reads the result -> no fp32 tensor is written, the
            # returned tensor is a storage-less autograd handle (the backward pass needs x and the statistics, not y)
            # (the handle takes the bf16 type under bf16 storage: the consumer's data gradient then comes back in bf16)
            y = (torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32) if keep_f32 else
                 torch.empty_strided(tuple(x.shape), (0, 0, 0, 0), device=x.device,
                                     dtype=torch.bfloat16 if bf16_handles_active() else torch.float32))
            check(L.jaf_layernorm_lrelu_fwd_packed_dt(_s(), _p(x), xb, _p(stats), _p(gamma), _p(beta), _p(y) if keep_f32 else None,
                                                      _p(dst.image.buf), dst.image.ng8, dst.coff, N, C, H * W, slope, _PRECISION),
                  "jaf_layernorm_lrelu_fwd_packed_dt")
        else:
            y = torch.empty(tuple(x.shape), device=x.device, dtype=torch.float32)
            check(L.jaf_layernorm_lrelu_fwd_dt(_s(), _p(x), xb, _p(stats), _p(gamma), _p(beta), _p(y), N, C, H * W, slope),
                  "jaf_layernorm_lrelu_fwd_dt")
        ctx.eps, ctx.slope = eps, slope
        ctx.mode = _bwd_mode()
        ctx.save_for_backward(x, gamma, beta, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        with _arith(ctx.mode):
            return _LayerNormLReLUFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        N, C, H, W = x.shape
        dy = _c(dy)
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        db16, xb16 = (1 if dy.dtype == torch.bfloat16 else 0), (1 if x.dtype == torch.bfloat16 else 0)
        # the kernel accumulates (+=): parameters that already own a .grad buffer are updated in place
        gi, bi = _grad_inplace(gamma), _grad_inplace(beta)
        dgamma = gamma.grad if gi else torch.zeros_like(gamma)
        dbeta = beta.grad if bi else torch.zeros_like(beta)
        ws = torch.empty(32 * N, device=x.device, dtype=torch.float64)      # [N][16 slots][2], include/jafpro_hip.h
        prod = getattr(ctx, "prod", None)
        if prod is not None and _packed_path_now() and prod.mode == ctx.mode:
            ctx.prod = None
            dzp = torch.empty(N * ((C + 7) // 8) * H * W * 16 * (2 if _PRECISION == PREC_BF16X3 else 1), device=x.device, dtype=torch.uint8)
            scratch = torch.empty(2 * N * C, device=x.device, dtype=torch.float32)
            pb = prod.bias_ref
            db, dbt, acc = None, None, 0
            if prod.has_bias and prod.needs_input_grad[1]:
                if _grad_inplace(pb):
                    dbt, acc = pb.grad, 1
                else:
                    dbt = db = torch.empty(C, device=x.device, dtype=torch.float32)
            with _hbm("ln_bwd_apply_packed_kernel", x.numel() * (dy.element_size() + x.element_size()) + dzp.numel()):
                check(lib().jaf_layernorm_lrelu_bwd_packed_dt(_s(), _p(dy), db16, _p(x), xb16, _p(stats), _p(gamma), _p(beta), _p(dzp), _p(dgamma),
                                                              _p(dbeta), _p(ws), _p(scratch), _p(dbt), acc, N, C, H * W, ctx.slope, ctx.eps,
                                                              _PRECISION), "jaf_layernorm_lrelu_bwd_packed_dt")
            prod.fused = (dzp, db)
            FUSED_STATS["ln"] += 1
            dx = torch.empty_strided(tuple(x.shape), (0, 0, 0, 0), device=x.device, dtype=x.dtype)     # placeholder
            return dx, (None if gi else dgamma), (None if bi else dbeta), None, None, None, None, None, None
        dx = torch.empty_like(x)
        check(lib().jaf_layernorm_lrelu_bwd_dt(_s(), _p(dy), db16, _p(x), xb16, _p(stats), _p(gamma), _p(beta), _p(dx), _p(dgamma),
                                               _p(dbeta), _p(ws), N, C, H * W, ctx.slope, ctx.eps),
              "jaf_layernorm_lrelu_bwd_dt")
        return dx, (None if gi else dgamma), (None if bi else dbeta), None, None, None, None, None, None


def layernorm_lrelu(x, gamma, beta, eps: float = 1e-5, slope: float = 0.01, pre: Optional[LNStats] = None,
                    dst: Optional[PackedDst] = None, keep_f32: bool = True, sole_consumer: bool = False):
    """`dst`: the consumer convolution's packed image slot; keep_f32=False: that convolution is the only reader, so no
    fp32 result is written (both on the packed bf16 path only, see PackedImage)."""
    _chk(x, "layernorm x", torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32); _chk(gamma, "gamma"); _chk(beta, "beta")
    return _LayerNormLReLUFn.apply(x, gamma, beta, eps, slope, pre, dst, keep_f32, sole_consumer)


class _BatchNormActFn(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training: bool, act: int, slope: float, residual,
                eps: float, momentum: float):
        N, C, H, W = x.shape
        L = lib()
        stats = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        ws = torch.empty(2 * C, device=x.device, dtype=torch.float64)
        y = torch.empty_like(x)
        check(L.jaf_batchnorm_act_fwd_fused(_s(), _p(x), N, C, H * W, eps, momentum, _p(running_mean), _p(running_var),
                                            _p(stats), 1 if training else 0, _p(ws), _p(weight), _p(bias), _p(residual), _p(y),
                                            act, slope), "jaf_batchnorm_act_fwd_fused")
        ctx.cfg = (training, act, slope, residual is not None)
        ctx.bias_ref = bias
        ctx.save_for_backward(x, y, weight, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, stats = ctx.saved_tensors
        training, act, slope, has_res = ctx.cfg
        N, C, H, W = x.shape
        dy = _c(dy)
        dx = torch.empty_like(x)
        bias = ctx.bias_ref
        inplace = _grad_inplace(weight) and bias is not None and _grad_inplace(bias)
        dw = weight.grad if inplace else torch.empty_like(weight)
        db = bias.grad if inplace else torch.empty_like(weight)
        ws = torch.empty(2 * C, device=x.device, dtype=torch.float64)
        check(lib().jaf_batchnorm_act_bwd(_s(), _p(dy), _p(x), _p(y), _p(stats), _p(weight), _p(dx), _p(dw), _p(db), N,
                                          C, H * W, act, slope, 1 if training else 0, _p(ws), 1 if inplace else 0),
              "jaf_batchnorm_act_bwd")
        return dx, (None if inplace else dw), (None if inplace else db), None, None, None, None, None, \
            (dy if has_res else None), None, None


_SPLIT_BN_ONE_LAUNCH = True          # (tests flip it: the per-chunk launches are the reference of the one-launch form)


class _SplitBatchNormActFn(Function):
    """BatchNorm (+ activation) of `parts` equal chunks of the batch, each with ITS OWN batch statistics, the running
    statistics updated chunk after chunk: what `parts` successive calls of the module on the chunks compute (the reference
    runs its discriminators on the real and on the generated images in separate calls, train/4...py:380-394), for a batch
    that went through the convolutions once.  The same kernels, called per chunk."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training: bool, act: int, slope: float, eps: float,
                momentum: float, parts: int):
        N, C, H, W = x.shape
        n = N // parts
        L = lib()
        stats = torch.empty(parts, 2 * C, device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        # all chunks in ONE launch where a chunk's channel fits a workgroup (bit-identical to the per-chunk calls below)
        rc = L.jaf_batchnorm_act_fwd_split(_s(), _p(x), N, C, H * W, eps, momentum, _p(running_mean), _p(running_var), _p(stats),
                                           _p(weight), _p(bias), _p(y), act, slope, parts) if (training and _SPLIT_BN_ONE_LAUNCH) else -2
        if rc not in (0, -2):
            check(rc, "jaf_batchnorm_act_fwd_split")
        ws = torch.empty(parts, 2 * C, device=x.device, dtype=torch.float64) if rc != 0 else None
        for k in range(parts if rc != 0 else 0):
            check(L.jaf_batchnorm_act_fwd_fused(_s(), _p(x[k * n:(k + 1) * n]), n, C, H * W, eps, momentum, _p(running_mean),
                                                _p(running_var), _p(stats[k]), 1 if training else 0, _p(ws[k]), _p(weight),
                                                _p(bias), None, _p(y[k * n:(k + 1) * n]), act, slope),
                  "jaf_batchnorm_act_fwd_fused")
        ctx.cfg = (training, act, slope, parts)
        ctx.bias_ref = bias
        ctx.save_for_backward(x, y, weight, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, stats = ctx.saved_tensors
        training, act, slope, parts = ctx.cfg
        N, C, H, W = x.shape
        n = N // parts
        dy = _c(dy)
        dx = torch.empty_like(x)
        bias = ctx.bias_ref
        inplace = _grad_inplace(weight) and bias is not None and _grad_inplace(bias)
        dw = weight.grad if inplace else torch.empty_like(weight)
        db = bias.grad if inplace else torch.empty_like(weight)
        rc = lib().jaf_batchnorm_act_bwd_split(_s(), _p(dy), _p(x), _p(y), _p(stats), _p(weight), _p(dx), _p(dw), _p(db), N, C, H * W, act,
                                               slope, 1 if training else 0, 1 if inplace else 0, parts) if _SPLIT_BN_ONE_LAUNCH else -2
        if rc not in (0, -2):
            check(rc, "jaf_batchnorm_act_bwd_split")
        ws = torch.empty(parts, 2 * C, device=x.device, dtype=torch.float64) if rc != 0 else None
        for k in range(parts if rc != 0 else 0):
            sl = slice(k * n, (k + 1) * n)
            check(lib().jaf_batchnorm_act_bwd(_s(), _p(dy[sl]), _p(x[sl]), _p(y[sl]), _p(stats[k]), _p(weight), _p(dx[sl]), _p(dw),
                                              _p(db), n, C, H * W, act, slope, 1 if training else 0, _p(ws[k]),
                                              1 if (inplace or k > 0) else 0), "jaf_batchnorm_act_bwd")
        return dx, (None if inplace else dw), (None if inplace else db), None, None, None, None, None, None, None, None


def batchnorm_act(x, weight, bias, running_mean, running_var, training=True, act=ACT_NONE, slope=0.0, residual=None,
                  eps=1e-5, momentum=0.1, batch_parts: int = 1):
    """`batch_parts` > 1: see _SplitBatchNormActFn (no residual)."""
    _chk(x, "batchnorm x")
    if batch_parts > 1:
        if residual is not None or x.shape[0] % batch_parts:
            raise RuntimeError("batchnorm_act: batch_parts needs a divisible batch and no residual")
        return _SplitBatchNormActFn.apply(x, weight, bias, running_mean, running_var, training, act, slope, eps, momentum,
                                          batch_parts)
    if residual is not None:
        _chk(residual, "residual")
    return _BatchNormActFn.apply(x, weight, bias, running_mean, running_var, training, act, slope, residual, eps,
                                 momentum)


# --------------------------------------------------------------------------------------------
# resampling
# --------------------------------------------------------------------------------------------
class _AvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        N, C, H, W = x.shape
        OH, OW = _out_size(H, k, stride, pad), _out_size(W, k, stride, pad)
        y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32)
        for n0, n1 in _n_chunks(N, C):
            check(lib().jaf_avgpool_fwd(_s(), _p(x[n0:n1]), _p(y[n0:n1]), (n1 - n0) * C, H, W, OH, OW, k, stride, pad), "jaf_avgpool_fwd")
        ctx.cfg = (N, C, H, W, OH, OW, k, stride, pad)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, OH, OW, k, stride, pad = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        for n0, n1 in _n_chunks(N, C):
            check(lib().jaf_avgpool_bwd(_s(), _p(dy[n0:n1]), _p(dx[n0:n1]), (n1 - n0) * C, H, W, OH, OW, k, stride, pad), "jaf_avgpool_bwd")
        return dx, None, None, None


def avg_pool(x, k: int, stride: int, pad: int):
    return _AvgPoolFn.apply(_chk(x, "avg_pool x"), k, stride, pad)


class _ResizeFn(Function):
    @staticmethod
    def forward(ctx, x, OH, OW, align, nearest, crop):
        N, C, H, W = x.shape
        y0, x0, ch, cw = crop if crop is not None else (0, 0, H, W)
        y = torch.empty((N, C, OH, OW), device=x.device, dtype=torch.float32)
        for n0, n1 in _n_chunks(N, C):
            check(lib().jaf_resize_fwd(_s(), _p(x[n0:n1]), _p(y[n0:n1]), n1 - n0, C, H, W, y0, x0, ch, cw, OH, OW, 1 if align else 0,
                                       1 if nearest else 0), "jaf_resize_fwd")
        ctx.cfg = (N, C, H, W, y0, x0, ch, cw, OH, OW, align, nearest)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, y0, x0, ch, cw, OH, OW, align, nearest = ctx.cfg
        if nearest:
            raise RuntimeError("nearest resize has no backward on this path")
        dy = _c(dy)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        for n0, n1 in _n_chunks(N, C):
            check(lib().jaf_resize_bwd(_s(), _p(dy[n0:n1]), _p(dx[n0:n1]), n1 - n0, C, H, W, y0, x0, ch, cw, OH, OW, 1 if align else 0),
                  "jaf_resize_bwd")
        return dx, None, None, None, None, None


class _LazyResizeFn(Function):
    """Bilinear resize whose result is only ever read by a convolution of the packed bf16 path: nothing is computed here,
    the consumer samples `x` while it packs its input image (pack_input -> jaf_conv2d_pack_input_resized).  The returned
    tensor is a storage-less autograd handle; the adjoint is the ordinary resize backward."""

    @staticmethod
    def forward(ctx, x, OH, OW, align):
        N, C, H, W = x.shape
        ctx.cfg = (N, C, H, W, OH, OW, align)
        # (bf16 storage: a bf16 handle, so that the consumer's data gradient -- OH*OW/(H*W) times the size of x -- arrives in bf16)
        return torch.empty_strided((N, C, OH, OW), (0, 0, 0, 0), device=x.device,
                                   dtype=torch.bfloat16 if bf16_handles_active() else torch.float32)

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, OH, OW, align = ctx.cfg
        dy = _c(dy)
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        for n0, n1 in _n_chunks(N, C):
            check(lib().jaf_resize_bwd_dt(_s(), _p(dy[n0:n1]), 1 if dy.dtype == torch.bfloat16 else 0, _p(dx[n0:n1]), n1 - n0, C, H, W, 0, 0, H, W,
                                          OH, OW, 1 if align else 0), "jaf_resize_bwd_dt")
        return dx, None, None, None


def resize(x, size: Tuple[int, int], align_corners: bool, nearest: bool = False, crop=None, lazy: bool = False):
    """Bilinear (or nearest) resize of x (optionally of the crop window (y0, x0, h, w)).
    lazy=True (honoured on the packed bf16 path, bilinear, no crop): the caller guarantees that the result is consumed
    only as a source of ops.conv2d, which then samples `x` itself while packing (see _LazyResizeFn)."""
    _chk(x, "resize x")
    if lazy and lazy_resize_active() and not nearest and crop is None:
        y = _LazyResizeFn.apply(x, int(size[0]), int(size[1]), bool(align_corners))
        y._jaf_lazy = (x, bool(align_corners))
        return y
    return _ResizeFn.apply(x, int(size[0]), int(size[1]), bool(align_corners), bool(nearest), crop)


class _ReflectPadFn(Function):
    @staticmethod
    def forward(ctx, x, p):
        N, C, H, W = x.shape
        y = torch.empty((N, C, H + 2 * p, W + 2 * p), device=x.device, dtype=torch.float32)
        for n0, n1 in _n_chunks(N, C):
            check(lib().jaf_reflect_pad_fwd(_s(), _p(x[n0:n1]), _p(y[n0:n1]), (n1 - n0) * C, H, W, p), "jaf_reflect_pad_fwd")
        ctx.cfg = (N, C, H, W, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, p = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32)
        for n0, n1 in _n_chunks(N, C):
            check(lib().jaf_reflect_pad_bwd(_s(), _p(dy[n0:n1]), _p(dx[n0:n1]), (n1 - n0) * C, H, W, p), "jaf_reflect_pad_bwd")
        return dx, None


def reflect_pad(x, p: int):
    return _ReflectPadFn.apply(_chk(x, "reflect_pad x"), p)


# --------------------------------------------------------------------------------------------
# gathers and blends
# --------------------------------------------------------------------------------------------
class _TextureWarpFn(Function):
    @staticmethod
    def forward(ctx, tex, iuv, align):
        B, C, TH, TW = tex.shape
        S = iuv.shape[1]
        out = torch.empty((B, 3, S, S), device=tex.device, dtype=torch.float32)
        with _hbm("texture_warp_fwd_kernel", B * S * S * (3.0 + 12.0 + 12.0)):    # IUV bytes + output + >= one texel/px
            check(lib().jaf_texture_warp_fwd(_s(), _p(tex), _p(iuv), _p(out), B, S, TH, TW, 1 if align else 0),
                  "jaf_texture_warp_fwd")
        ctx.cfg = (B, S, TH, TW, align)
        ctx.save_for_backward(iuv)
        return out

    @staticmethod
    def backward(ctx, dout):
        (iuv,) = ctx.saved_tensors
        B, S, TH, TW, align = ctx.cfg
        dout = _c(dout)
        dtex = torch.zeros((B, 72, TH, TW), device=dout.device, dtype=torch.float32)
        with _hbm("texture_warp_bwd_kernel", B * S * S * (3.0 + 12.0 + 12.0)):
            check(lib().jaf_texture_warp_bwd(_s(), _p(dout), _p(iuv), _p(dtex), B, S, TH, TW, 1 if align else 0),
                  "jaf_texture_warp_bwd")
        return dtex, None, None


def texture_warp(tex: torch.Tensor, iuv255: torch.Tensor, align_corners: bool = False):
    """tex [B, 72, 200, 200] (24 parts x 3), iuv255 uint8 [B, S, S, 3] -> [B, 3, S, S]."""
    _chk(tex, "texture_warp tex")
    _chk(iuv255, "texture_warp iuv", torch.uint8)
    if tex.shape[1] != 72 or iuv255.dim() != 4 or iuv255.shape[3] != 3 or iuv255.shape[1] != iuv255.shape[2]:
        raise RuntimeError("texture_warp: bad shapes %s %s" % (tuple(tex.shape), tuple(iuv255.shape)))
    return _TextureWarpFn.apply(tex, iuv255, align_corners)


class _GridSampleFn(Function):
    @staticmethod
    def forward(ctx, src, grid, padding_border, align_corners):
        B, C, H, W = src.shape
        OH, OW = grid.shape[1], grid.shape[2]
        out = torch.empty((B, C, OH, OW), device=src.device, dtype=torch.float32)
        with _hbm("grid_sample_fwd_kernel", 4.0 * B * (C * H * W + 2 * OH * OW + C * OH * OW)):
            check(lib().jaf_grid_sample_fwd(_s(), _p(src), _p(grid), _p(out), B, C, H, W, OH, OW, 1 if padding_border else 0,
                                            1 if align_corners else 0), "jaf_grid_sample_fwd")
        ctx.cfg = (padding_border, align_corners)
        ctx.save_for_backward(src, grid)
        return out

    @staticmethod
    def backward(ctx, dout):
        src, grid = ctx.saved_tensors
        padding_border, align_corners = ctx.cfg
        B, C, H, W = src.shape
        OH, OW = grid.shape[1], grid.shape[2]
        dout = _c(dout)
        dsrc = torch.zeros_like(src) if ctx.needs_input_grad[0] else None
        dgrid = torch.empty_like(grid) if ctx.needs_input_grad[1] else None
        check(lib().jaf_grid_sample_bwd(_s(), _p(dout), _p(src), _p(grid), _p(dsrc), _p(dgrid), B, C, H, W, OH, OW,
                                        1 if padding_border else 0, 1 if align_corners else 0), "jaf_grid_sample_bwd")
        return dsrc, dgrid, None, None


def grid_sample(src, grid, padding_border: bool, align_corners: bool = False):
    """Bilinear F.grid_sample (zeros or border padding).  Stage 4 only runs it forward (the flow warp carries no
    gradient there); the adjoint w.r.t. the image and the grid is the differentiable flow of SURVEY 8(f1)."""
    _chk(src, "grid_sample src"); _chk(grid, "grid_sample grid")
    if src.requires_grad or grid.requires_grad:
        return _GridSampleFn.apply(src, grid, bool(padding_border), bool(align_corners))
    with torch.no_grad():
        return _GridSampleFn.apply(src, grid, bool(padding_border), bool(align_corners))


class _BlendFn(Function):
    @staticmethod
    def forward(ctx, a, b, m):
        N, C, H, W = a.shape
        out = torch.empty_like(a)
        with _hbm("blend_fwd_kernel", 4.0 * N * H * W * (3 * C + 1)):
            check(lib().jaf_blend_fwd(_s(), _p(a), _p(b), _p(m), _p(out), N, C, H * W), "jaf_blend_fwd")
        ctx.save_for_backward(a, b, m)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, m = ctx.saved_tensors
        N, C, H, W = a.shape
        dout = _c(dout)
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        dm = torch.empty_like(m) if ctx.needs_input_grad[2] else None
        nout = (C if da is not None else 0) + (C if db is not None else 0) + (1 if dm is not None else 0)
        with _hbm("blend_bwd_kernel", 4.0 * N * H * W * (3 * C + 1 + nout)):
            check(lib().jaf_blend_bwd(_s(), _p(dout), _p(a), _p(b), _p(m), _p(da), _p(db), _p(dm), N, C, H * W),
                  "jaf_blend_bwd")
        return da, db, dm


def blend(a, b, m):
    """a*m + b*(1-m); m is [N,1,H,W]."""
    return _BlendFn.apply(_chk(a, "blend a"), _chk(b, "blend b"), _chk(m, "blend m"))


class _MulBcastFn(Function):
    @staticmethod
    def forward(ctx, x, m):
        N, C, H, W = x.shape
        out = torch.empty_like(x)
        check(lib().jaf_mul_bcast(_s(), _p(x), _p(m), _p(out), N, C, m.shape[1], H * W), "jaf_mul_bcast")
        ctx.save_for_backward(m)
        return out

    @staticmethod
    def backward(ctx, dout):
        (m,) = ctx.saved_tensors
        dout = _c(dout)
        N, C, H, W = dout.shape
        dx = torch.empty_like(dout)
        check(lib().jaf_mul_bcast(_s(), _p(dout), _p(m), _p(dx), N, C, m.shape[1], H * W), "jaf_mul_bcast")
        return dx, None


def mul_bcast(x, m):
    """x * m with m [N,1,H,W] or [N,C,H,W] (m carries no gradient: masks are data)."""
    return _MulBcastFn.apply(_chk(x, "mul x"), _chk(m, "mul m"))


class _PartMaskMulFn(Function):
    @staticmethod
    def forward(ctx, tex, masks, used):
        B, PC, PS, _ = tex.shape
        T, AH, AW = masks.shape[1], masks.shape[2], masks.shape[3]
        out = torch.empty_like(tex)
        check(lib().jaf_part_mask_mul(_s(), _p(tex), _p(masks), _p(used), _p(out), B, T, AH, AW, PC // 3, PS),
              "jaf_part_mask_mul")
        ctx.save_for_backward(masks, used)
        return out

    @staticmethod
    def backward(ctx, dout):
        masks, used = ctx.saved_tensors
        dout = _c(dout)
        B, PC, PS, _ = dout.shape
        T, AH, AW = masks.shape[1], masks.shape[2], masks.shape[3]
        dx = torch.empty_like(dout)
        check(lib().jaf_part_mask_mul(_s(), _p(dout), _p(masks), _p(used), _p(dx), B, T, AH, AW, PC // 3, PS),
              "jaf_part_mask_mul")
        return dx, None, None


def part_mask_mul(tex, masks, used):
    """tex [B,72,200,200] * OR_t(masks[b,t] & used[t]) cut per part (train/4...py:283-298)."""
    _chk(tex, "tex"); _chk(masks, "masks"); _chk(used, "used", torch.int32)
    return _PartMaskMulFn.apply(tex, masks, used)


def atlas_to_parts(atlas: torch.Tensor, psz: int = 200) -> torch.Tensor:
    """atlas [B,T,3,AH,AW] -> [T*B, 24*3, psz, psz] (image t*B+b), train/4...py:269-276."""
    _chk(atlas, "atlas")
    B, T, _, AH, AW = atlas.shape
    P = (AH // psz) * (AW // psz)
    out = torch.empty((T * B, P * 3, psz, psz), device=atlas.device, dtype=torch.float32)
    check(lib().jaf_atlas_to_parts(_s(), _p(atlas), _p(out), B, T, AH, AW, psz), "jaf_atlas_to_parts")
    return out


def atlas_to_parts_packed(atlas: torch.Tensor, psz: int = 200):
    """The same slicing as the packed bf16 input image of the first part-encoder convolution: returns (x, image) where `image` is
    the PackedImage [T*B][24][3 -> 8 channels] and `x` a storage-less [T*B, 72, psz, psz] tensor that only carries the shape
    (conv2d(..., prepacked=image) reads nothing else).  None while the packed path is off or the geometry is not 16-byte
    friendly: the caller then takes atlas_to_parts + the packing pass."""
    _chk(atlas, "atlas")
    if not packed_active():
        return None
    B, T, _, AH, AW = atlas.shape
    P = (AH // psz) * (AW // psz)
    img = PackedImage(T * B, P, 3, psz, psz, atlas.device)
    rc = lib().jaf_atlas_to_parts_packed(_s(), _p(atlas), _p(img.buf), B, T, AH, AW, psz, _PRECISION)
    if rc == -2:            # JAF_EUNSUPPORTED
        return None
    check(rc, "jaf_atlas_to_parts_packed")
    x = torch.empty_strided((T * B, P * 3, psz, psz), (0, 0, 0, 0), device=atlas.device, dtype=torch.float32)
    return x, img


# --------------------------------------------------------------------------------------------
# losses / classifier / optimiser
# --------------------------------------------------------------------------------------------
class _VggPreFn(Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        check(lib().jaf_vgg_preprocess(_s(), _p(x), _p(y), N, H * W), "jaf_vgg_preprocess")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(lib().jaf_axpby(_s(), 127.5, _p(dy), 0.0, _p(dx), dy.numel()), "jaf_axpby")
        return dx


def vgg_preprocess(x):
    x = _chk(x, "vgg_preprocess x")
    if x.shape[1] != 3:
        raise RuntimeError("vgg_preprocess expects 3 channels")
    return _VggPreFn.apply(x)


class _L1Fn(Function):
    @staticmethod
    def forward(ctx, a, b, w):
        loss = torch.zeros(1, device=a.device, dtype=torch.float32)
        check(lib().jaf_l1_loss_fwd(_s(), _p(a), _p(b), a.numel(), w, _p(loss)), "jaf_l1_loss_fwd")
        ctx.w = w
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        a, b = ctx.saved_tensors
        dloss = _c(dloss)
        da = torch.empty_like(a)
        check(lib().jaf_l1_loss_bwd(_s(), _p(a), _p(b), a.numel(), ctx.w, _p(dloss), _p(da), 0), "jaf_l1_loss_bwd")
        return da, None, None


def l1_loss(a, b, weight: float = 1.0):
    """weight * mean|a-b| as a 1-element tensor; gradient flows to `a` only (b is the detached
    target branch, src/networks.py:106)."""
    _chk(a, "l1 a"); _chk(b, "l1 b")
    if a.shape != b.shape:
        raise RuntimeError("l1_loss shape mismatch")
    return _L1Fn.apply(a, b, float(weight))


class _BCEFn(Function):
    @staticmethod
    def forward(ctx, p, target):
        loss = torch.empty(1, device=p.device, dtype=torch.float32)
        check(lib().jaf_bce_fwd(_s(), _p(p), p.numel(), target, _p(loss)), "jaf_bce_fwd")
        ctx.target = target
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (p,) = ctx.saved_tensors
        dloss = _c(dloss)
        dp = torch.empty_like(p)
        check(lib().jaf_bce_bwd(_s(), _p(p), p.numel(), ctx.target, _p(dloss), _p(dp)), "jaf_bce_bwd")
        return dp, None


def bce_loss(p, target: float):
    """nn.BCELoss()(p, full_like(p, target)) -> 1-element tensor."""
    return _BCEFn.apply(_chk(p, "bce p"), float(target))


class _BCEPairFn(Function):
    @staticmethod
    def forward(ctx, p, n1: int, t1: float, t2: float):
        l1 = torch.empty(1, device=p.device, dtype=torch.float32)
        l2 = torch.empty(1, device=p.device, dtype=torch.float32)
        ls = torch.empty(1, device=p.device, dtype=torch.float32)
        check(lib().jaf_bce_pair_fwd(_s(), _p(p), n1, p.numel(), t1, t2, _p(l1), _p(l2), _p(ls)), "jaf_bce_pair_fwd")
        ctx.cfg = (n1, t1, t2)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(p)
        return l1, l2, ls

    @staticmethod
    def backward(ctx, g1, g2, gs):
        (p,) = ctx.saved_tensors
        n1, t1, t2 = ctx.cfg
        dp = torch.empty_like(p)
        c = lambda t: None if t is None else _p(_c(t))
        check(lib().jaf_bce_pair_bwd(_s(), _p(p), n1, p.numel(), t1, t2, c(g1), c(g2), c(gs), _p(dp)), "jaf_bce_pair_bwd")
        return dp, None, None, None


def bce_pair(p, n1: int, t1: float = 1.0, t2: float = 0.0):
    """(BCE(p[:n1], t1), BCE(p[n1:], t2), their sum) of one probability vector in one launch each way (the discriminators' real and
    generated halves of a batched pass)."""
    p = _chk(p, "bce_pair p")
    if not (1 <= n1 < p.numel()):
        raise RuntimeError("bce_pair: n1 must split the %d probabilities" % p.numel())
    return _BCEPairFn.apply(p, int(n1), float(t1), float(t2))


_SEED_ONE: dict = {}


def backward_from(loss: torch.Tensor) -> None:
    """loss.backward() with a cached one as the seed gradient (autograd otherwise launches a fill for it on every call)."""
    one = _SEED_ONE.get(loss.device)
    if one is None:
        one = _SEED_ONE[loss.device] = torch.ones(1, device=loss.device, dtype=torch.float32)
    torch.autograd.backward([loss], [one.view_as(loss)])


_LINEAR_FUSED_BWD = True          # (tests flip it)


class _LinearFn(Function):
    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        N, I = x.shape
        O = w.shape[0]
        y = torch.empty((N, O), device=x.device, dtype=torch.float32)
        check(lib().jaf_linear_fwd(_s(), _p(x), _p(w), _p(b), _p(y), N, I, O, act, slope), "jaf_linear_fwd")
        ctx.cfg = (act, slope)
        ctx.bias_ref = b
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        b = ctx.bias_ref
        if _LINEAR_FUSED_BWD and b is not None and _grad_inplace(w) and _grad_inplace(b):
            return _LinearFn._backward_fused(ctx, dy)
        act, slope = ctx.cfg
        N, I = x.shape
        O = w.shape[0]
        dy = _c(dy)
        L = lib()
        if act != ACT_NONE:
            dz = torch.empty_like(dy)
            check(L.jaf_act_bwd(_s(), _p(dy), _p(y), _p(dz), dy.numel(), act, slope), "jaf_act_bwd")
        else:
            dz = dy
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(O, device=x.device, dtype=torch.float32)
        check(L.jaf_linear_bwd(_s(), _p(dz), _p(x), _p(w), _p(dx), _p(dw), _p(db), N, I, O), "jaf_linear_bwd")
        return dx, dw, db, None, None

    @staticmethod
    def _backward_fused(ctx, dy):
        # activation backward inside the kernel, parameter gradients added to their .grad buffers: one launch
        x, w, y = ctx.saved_tensors
        act, slope = ctx.cfg
        N, I = x.shape
        O = w.shape[0]
        b = ctx.bias_ref
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        check(lib().jaf_linear_bwd_fused(_s(), _p(_c(dy)), _p(y), _p(x), _p(w), _p(dx), _p(w.grad), _p(b.grad), N, I, O, act, slope, 1),
              "jaf_linear_bwd_fused")
        return dx, None, None, None, None


def linear(x, w, b, act=ACT_NONE, slope=0.0):
    return _LinearFn.apply(_chk(x, "linear x"), _chk(w, "linear w"), _chk(b, "linear b"), act, float(slope))


def adam_step(p, g, m, v, lr: float, step: int, beta1=0.9, beta2=0.999, eps=1e-8, params=None, refresh: bool = False,
              state: Optional[torch.Tensor] = None):
    """torch.optim.Adam defaults over one flat buffer.  `state` (fp32 [4] on the device, element 0 = the int32 count of steps
    taken so far): the step count lives on the device and `step` is ignored -- the form a captured hipGraph needs."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, "adam " + n)
    if state is not None:
        check(lib().jaf_adam_step_dev(_s(), _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, _p(_chk(state, "adam state"))),
              "jaf_adam_step_dev")
    else:
        check(lib().jaf_adam_step(_s(), _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step),
              "jaf_adam_step")
    if refresh and params is not None:
        refresh_packed_weights(params)
    else:
        invalidate_packed_weights(params)


def reset_pack_events():
    """Forgets the cross-stream events of the cached weight images (everything is idle: call after a device synchronise).
    Events recorded while a hipGraph was being captured are not real events, and a capture must not wait on events recorded
    before it began (GraphedTrainStep brackets its capture with this)."""
    for e in _PACK_CACHE.values():
        e.event, e.waited = None, set()


def cache_census():
    """Sizes of the host-side caches a train step may still be filling (plans, descriptors, weight images, re-pack tables)."""
    return (len(_PLAN_CACHE), len(_DESC_CACHE), len(_PACK_CACHE), len(_REPACK_TABLES), len(_SMALL_CONSTS))


def pack_cache_entries():
    return list(_PACK_CACHE.values())


def invalidate_packed_weights(params=None):
    """Weights written behind torch's back (the Adam kernel, RCCL broadcasts) do not bump
    tensor._version, so their packed images are dropped explicitly -- only those of `params` when given
    (frozen networks such as the background CRN and VGG keep theirs across steps)."""
    if params is None:
        _PACK_CACHE.clear()
        _PACK_KEYS_BY_ID.clear()
        return
    for t in params:
        for ck in _PACK_KEYS_BY_ID.pop(id(t), ()):
            _PACK_CACHE.pop(ck, None)


def axpby(a: float, x, b: float, y):
    """y = a*x + b*y in place."""
    check(lib().jaf_axpby(_s(), a, _p(_chk(x, "x")), b, _p(_chk(y, "y")), x.numel()), "jaf_axpby")
    return y


# --------------------------------------------------------------------------------------------
# renderer path (forward only: vertices carry no gradient in stage 4)
# --------------------------------------------------------------------------------------------
class _ProjectFacesFn(Function):
    @staticmethod
    def forward(ctx, verts, cam, faces_idx, eye_z):
        B, NV, _ = verts.shape
        NF = faces_idx.shape[0]
        out = torch.empty((B, NF, 3, 3), device=verts.device, dtype=torch.float32)
        check(lib().jaf_project_faces(_s(), _p(verts), _p(cam), _p(faces_idx), _p(out), B, NV, NF, eye_z),
              "jaf_project_faces")
        ctx.save_for_backward(verts, cam, faces_idx)
        return out

    @staticmethod
    def backward(ctx, dfaces):
        verts, cam, faces_idx = ctx.saved_tensors
        B, NV, _ = verts.shape
        NF = faces_idx.shape[0]
        dfaces = _c(dfaces)
        dverts = torch.zeros_like(verts)
        dcam = torch.zeros_like(cam) if ctx.needs_input_grad[1] else None
        check(lib().jaf_project_faces_bwd(_s(), _p(dfaces), _p(verts), _p(cam), _p(faces_idx), _p(dverts), _p(dcam), B, NV, NF),
              "jaf_project_faces_bwd")
        return dverts, dcam, None, None


def project_faces(verts, cam, faces_idx, eye_z: float):
    """src/nmr.py:269-276 in one kernel: orthographic projection, y flip, look_at translation, vertices_to_faces.
    Differentiable w.r.t. the vertices and the camera."""
    _chk(verts, "verts"); _chk(cam, "cam"); _chk(faces_idx, "faces", torch.int32)
    return _ProjectFacesFn.apply(verts, cam, faces_idx, float(eye_z))


class _RasterizeFn(Function):
    """RasterizeFunction of neural_renderer (rasterize.py:16-160) without its texture branch: forward fills the
    face-index / weight / depth / face-inverse / alpha maps, backward is backward_pixel_map (alpha) followed by
    backward_depth_map.  Maps are UNFLIPPED here, like the reference Function's; the wrappers below flip."""

    @staticmethod
    def forward(ctx, faces, image_size, near, far, eps, return_alpha, return_depth):
        B, NF = faces.shape[0], faces.shape[1]
        S = image_size
        L = lib()
        dev = faces.device
        ws = torch.empty(int(L.jaf_rasterize_workspace(B, NF, S)), device=dev, dtype=torch.uint8)
        fim = torch.empty((B, S, S), device=dev, dtype=torch.int32)
        wim = torch.empty((B, S, S, 3), device=dev, dtype=torch.float32)
        depth = torch.empty((B, S, S), device=dev, dtype=torch.float32)
        finv = torch.empty((B, S, S, 3, 3), device=dev, dtype=torch.float32) if return_depth else None
        alpha = torch.empty((B, S, S), device=dev, dtype=torch.float32)
        check(L.jaf_rasterize_maps(_s(), _p(faces), _p(fim), _p(wim), _p(depth), _p(finv), _p(alpha), _p(ws), B, NF, S,
                                   near, far, 0), "jaf_rasterize_maps")
        ctx.cfg = (S, eps, return_alpha, return_depth)
        ctx.save_for_backward(faces, fim, wim, depth, finv, alpha)
        ctx.mark_non_differentiable(fim, wim)
        ctx.set_materialize_grads(False)
        return alpha.clone(), depth.clone(), fim, wim

    @staticmethod
    def backward(ctx, g_alpha, g_depth, _g_fim, _g_wim):
        faces, fim, wim, depth, finv, alpha = ctx.saved_tensors
        S, eps, return_alpha, return_depth = ctx.cfg
        B, NF = faces.shape[0], faces.shape[1]
        L = lib()
        g = torch.zeros_like(faces)
        if return_alpha and g_alpha is not None:
            check(L.jaf_rasterize_bwd_pixel_map(_s(), _p(faces), _p(fim), None, _p(alpha), None, _p(_c(g_alpha)), _p(g), B, NF, S,
                                                eps), "jaf_rasterize_bwd_pixel_map")
        if return_depth and g_depth is not None:
            check(L.jaf_rasterize_bwd_depth_map(_s(), _p(faces), _p(depth), _p(fim), _p(finv), _p(wim), _p(_c(g_depth)), _p(g),
                                                B, NF, S), "jaf_rasterize_bwd_depth_map")
        return g, None, None, None, None, None, None


def rasterize(faces, image_size: int, near: float = 0.1, far: float = 100.0, eps: float = 1e-4, return_alpha: bool = True,
              return_depth: bool = False):
    """-> (alpha, depth, fim, wim), unflipped (see _RasterizeFn); differentiable w.r.t. `faces` through alpha / depth."""
    _chk(faces, "faces")
    return _RasterizeFn.apply(faces, int(image_size), float(near), float(far), float(eps), bool(return_alpha), bool(return_depth))


def rasterize_silhouettes(faces, image_size: int, near: float = 0.1, far: float = 100.0, eps: float = 1e-4):
    """neural_renderer.rasterize_silhouettes without anti-aliasing (rasterize.py:428-452): alpha [B,S,S], flipped (:334-338)."""
    return torch.flip(rasterize(faces, image_size, near, far, eps, True, False)[0], dims=(1,))


def rasterize_depth(faces, image_size: int, near: float = 0.1, far: float = 100.0, eps: float = 1e-4):
    """neural_renderer.rasterize_depth without anti-aliasing (rasterize.py:455-481): depth [B,S,S], flipped."""
    return torch.flip(rasterize(faces, image_size, near, far, eps, False, True)[1], dims=(1,))


_SMALL_CONSTS = {}


def _dev_const(values, device) -> torch.Tensor:
    """A tiny fp32 constant (background colour ...) resident on the device; made once per (values, device)."""
    t = torch.as_tensor(values, dtype=torch.float32).reshape(-1) if not isinstance(values, torch.Tensor) else None
    if t is None:
        return _c(values.to(device=device, dtype=torch.float32))
    k = (tuple(float(v) for v in t), str(device))
    hit = _SMALL_CONSTS.get(k)
    if hit is None:
        hit = _SMALL_CONSTS[k] = t.to(device)
    return hit


class _RasterizeRGBFn(Function):
    """RasterizeFunction with return_rgb (rasterize.py:23-160): face-index map -> texture sampling -> background; backward
    = backward_pixel_map over (rgb [, alpha]) + backward_textures [+ backward_depth_map].  The two per-pixel sampling maps
    the reference saves (64 B/pixel each way) are not kept: the texture adjoint rebuilds its taps from the weight / depth
    maps (jaf_rasterize_texture_bwd_rebuild).  Maps UNFLIPPED; rgb [B,S,S,3]."""

    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background, return_alpha, return_depth):
        B, NF = faces.shape[0], faces.shape[1]
        S, ts = image_size, int(textures.shape[2])
        L, dev = lib(), faces.device
        ws = torch.empty(int(L.jaf_rasterize_workspace(B, NF, S)), device=dev, dtype=torch.uint8)
        fim = torch.empty((B, S, S), device=dev, dtype=torch.int32)
        wim = torch.empty((B, S, S, 3), device=dev, dtype=torch.float32)
        depth = torch.empty((B, S, S), device=dev, dtype=torch.float32)
        finv = torch.empty((B, S, S, 3, 3), device=dev, dtype=torch.float32) if return_depth else None
        alpha = torch.empty((B, S, S), device=dev, dtype=torch.float32)
        check(L.jaf_rasterize_maps(_s(), _p(faces), _p(fim), _p(wim), _p(depth), _p(finv), _p(alpha), _p(ws), B, NF, S,
                                   near, far, 0), "jaf_rasterize_maps")
        rgb = torch.empty((B, S, S, 3), device=dev, dtype=torch.float32)
        bg = _dev_const(background, dev)
        if bg.numel() not in (3, 3 * B):
            raise RuntimeError("background colour must be [3] or [B, 3]")
        check(L.jaf_rasterize_texture_fwd(_s(), _p(faces), _p(textures), _p(fim), _p(wim), _p(depth), _p(rgb), None, None, _p(bg),
                                          1 if (bg.numel() == 3 * B and B > 1) else 0, B, NF, S, ts, eps),
              "jaf_rasterize_texture_fwd")
        ctx.cfg = (S, ts, eps, return_alpha, return_depth)
        ctx.save_for_backward(faces, fim, wim, depth, finv, alpha, rgb)
        ctx.mark_non_differentiable(fim, wim)
        ctx.set_materialize_grads(False)
        return rgb.clone(), alpha.clone(), depth.clone(), fim, wim

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, _g_fim, _g_wim):
        faces, fim, wim, depth, finv, alpha, rgb = ctx.saved_tensors
        S, ts, eps, return_alpha, return_depth = ctx.cfg
        B, NF = faces.shape[0], faces.shape[1]
        L = lib()
        g = torch.zeros_like(faces)
        grgb = torch.zeros_like(rgb) if g_rgb is None else _c(g_rgb)
        ga = None
        if return_alpha:
            ga = torch.zeros_like(alpha) if g_alpha is None else _c(g_alpha)
        check(L.jaf_rasterize_bwd_pixel_map(_s(), _p(faces), _p(fim), _p(rgb), _p(alpha) if return_alpha else None, _p(grgb),
                                            _p(ga), _p(g), B, NF, S, eps), "jaf_rasterize_bwd_pixel_map")
        gt = None
        if ctx.needs_input_grad[1]:
            gt = torch.zeros((B, NF, ts, ts, ts, 3), device=faces.device, dtype=torch.float32)
            check(L.jaf_rasterize_texture_bwd_rebuild(_s(), _p(faces), _p(fim), _p(wim), _p(depth), _p(grgb), _p(gt), B, NF, S, ts,
                                                      eps), "jaf_rasterize_texture_bwd_rebuild")
        if return_depth and g_depth is not None:
            check(L.jaf_rasterize_bwd_depth_map(_s(), _p(faces), _p(depth), _p(fim), _p(finv), _p(wim), _p(_c(g_depth)), _p(g),
                                                B, NF, S), "jaf_rasterize_bwd_depth_map")
        return g, gt, None, None, None, None, None, None, None


def rasterize_rgb(faces, textures, image_size: int, near: float = 0.1, far: float = 100.0, eps: float = 1e-4,
                  background=(0.0, 0.0, 0.0), return_alpha: bool = False, return_depth: bool = False):
    """-> (rgb [B,S,S,3], alpha, depth, fim, wim), unflipped; differentiable w.r.t. `faces` and `textures` [B,NF,ts,ts,ts,3]."""
    _chk(faces, "faces"); _chk(textures, "textures")
    if textures.dim() != 6 or textures.shape[:2] != faces.shape[:2] or textures.shape[-1] != 3 or not (
            textures.shape[2] == textures.shape[3] == textures.shape[4]):
        raise RuntimeError("textures must be [B, NF, ts, ts, ts, 3]")
    return _RasterizeRGBFn.apply(faces, textures, int(image_size), float(near), float(far), float(eps), background,
                                 bool(return_alpha), bool(return_depth))


def rasterize_textured(faces, textures, image_size: int = 256, anti_aliasing: bool = True, near: float = 0.1, far: float = 100.0,
                       eps: float = 1e-4, background=(0.0, 0.0, 0.0)):
    """neural_renderer.rasterize (rasterize.py:361-391 -> rasterize_rgbad :257-358): RGB [B,3,S,S], vertically flipped,
    rendered at 2x and average-pooled when anti_aliasing."""
    S = image_size * 2 if anti_aliasing else image_size
    rgb = rasterize_rgb(faces, textures, S, near, far, eps, background)[0]
    rgb = torch.flip(rgb.permute(0, 3, 1, 2), dims=(2,)).contiguous()
    return avg_pool(rgb, 2, 2, 0) if anti_aliasing else rgb


def _f3(v):
    vals = [float(x) for x in (v.reshape(-1).tolist() if isinstance(v, torch.Tensor) else list(v))]
    if len(vals) != 3:
        raise RuntimeError("lighting: per-image colours / directions are not supported (expected 3 values)")
    return (ctypes.c_float * 3)(*vals)


class _LightingFn(Function):
    @staticmethod
    def forward(ctx, faces, textures, ia, idr, ca, cd, direction):
        B, NF, ts = faces.shape[0], faces.shape[1], int(textures.shape[2])
        out = torch.empty_like(textures)
        ctx.args = (float(ia), float(idr), _f3(ca), _f3(cd), _f3(direction), B, NF, ts)
        a = ctx.args
        check(lib().jaf_lighting_fwd(_s(), _p(faces), _p(textures), _p(out), None, a[0], a[1], a[2], a[3], a[4], B, NF, ts),
              "jaf_lighting_fwd")
        ctx.save_for_backward(faces, textures)
        return out

    @staticmethod
    def backward(ctx, g):
        faces, textures = ctx.saved_tensors
        ia, idr, ca, cd, direction, B, NF, ts = ctx.args
        g = _c(g)
        gt = torch.empty_like(textures) if ctx.needs_input_grad[1] else None
        gf = torch.empty_like(faces) if ctx.needs_input_grad[0] else None
        if gt is None and gf is None:
            return (None,) * 7
        check(lib().jaf_lighting_bwd(_s(), _p(faces), _p(textures), _p(g), _p(gt), _p(gf), ia, idr, ca, cd, direction, B, NF, ts),
              "jaf_lighting_bwd")
        return gf, gt, None, None, None, None, None


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """neural_renderer.lighting (lighting.py:6-58), out of place: textures [B,NF,ts,ts,ts,3] * light(faces [B,NF,3,3])."""
    _chk(faces, "faces"); _chk(textures, "textures")
    return _LightingFn.apply(faces, textures, intensity_ambient, intensity_directional, color_ambient, color_directional, direction)


class _VerticesToFacesFn(Function):
    @staticmethod
    def forward(ctx, verts, faces_idx):
        B, NV, NF = verts.shape[0], verts.shape[1], faces_idx.shape[0]
        out = torch.empty((B, NF, 3, 3), device=verts.device, dtype=torch.float32)
        check(lib().jaf_vertices_to_faces(_s(), _p(verts), _p(faces_idx), _p(out), B, NV, NF), "jaf_vertices_to_faces")
        ctx.save_for_backward(faces_idx)
        ctx.nv = NV
        return out

    @staticmethod
    def backward(ctx, g):
        (faces_idx,) = ctx.saved_tensors
        B, NF = g.shape[0], g.shape[1]
        dv = torch.zeros((B, ctx.nv, 3), device=g.device, dtype=torch.float32)
        check(lib().jaf_vertices_to_faces_bwd(_s(), _p(_c(g)), _p(faces_idx), _p(dv), B, ctx.nv, NF), "jaf_vertices_to_faces_bwd")
        return dv, None


def vertices_to_faces(verts, faces_idx):
    """neural_renderer.vertices_to_faces for one shared topology: verts [B,NV,3], faces_idx int32 [NF,3] -> [B,NF,3,3]."""
    _chk(verts, "verts"); _chk(faces_idx, "faces", torch.int32)
    return _VerticesToFacesFn.apply(verts, faces_idx)


class _FaceSamplerFn(Function):
    @staticmethod
    def forward(ctx, verts, cam, faces_idx, coords):
        B, NV, NF, TT = verts.shape[0], verts.shape[1], faces_idx.shape[0], coords.shape[1]
        out = torch.empty((B, NF, TT, 2), device=verts.device, dtype=torch.float32)
        check(lib().jaf_face_sampler_fwd(_s(), _p(verts), _p(cam), _p(faces_idx), _p(coords), _p(out), B, NV, NF, TT),
              "jaf_face_sampler_fwd")
        ctx.save_for_backward(verts, cam, faces_idx, coords)
        return out

    @staticmethod
    def backward(ctx, g):
        verts, cam, faces_idx, coords = ctx.saved_tensors
        B, NV, NF, TT = verts.shape[0], verts.shape[1], faces_idx.shape[0], coords.shape[1]
        dverts = torch.zeros_like(verts)
        dcam = torch.zeros_like(cam) if ctx.needs_input_grad[1] else None
        check(lib().jaf_face_sampler_bwd(_s(), _p(verts), _p(cam), _p(faces_idx), _p(coords), _p(_c(g)), _p(dverts), _p(dcam),
                                         B, NV, NF, TT), "jaf_face_sampler_bwd")
        return dverts, dcam, None, None


def face_sampler(verts, cam, faces_idx, coords):
    """SMPLRenderer.dynamic_sampler (src/nmr.py:388-395): [B,NF,T*T,2] image positions of every face's texels."""
    _chk(verts, "verts"); _chk(cam, "cam"); _chk(faces_idx, "faces", torch.int32); _chk(coords, "coords")
    return _FaceSamplerFn.apply(verts, cam, faces_idx, coords)


class _TexExpandFn(Function):
    @staticmethod
    def forward(ctx, sampled, T):
        B, _, NF, TT = sampled.shape
        out = torch.empty((B, NF, T, T, T, 3), device=sampled.device, dtype=torch.float32)
        check(lib().jaf_tex_expand_fwd(_s(), _p(sampled), _p(out), B, NF, T), "jaf_tex_expand_fwd")
        ctx.cfg = (B, NF, T)
        return out

    @staticmethod
    def backward(ctx, g):
        B, NF, T = ctx.cfg
        out = torch.empty((B, 3, NF, T * T), device=g.device, dtype=torch.float32)
        check(lib().jaf_tex_expand_bwd(_s(), _p(_c(g)), _p(out), B, NF, T), "jaf_tex_expand_bwd")
        return out, None


def tex_expand(sampled, T: int):
    """[B,3,NF,T*T] -> [B,NF,T,T,T,3] (view / permute / unsqueeze / repeat of SMPLRenderer.extract_tex, src/nmr.py:379-384)."""
    _chk(sampled, "sampled")
    if sampled.dim() != 4 or sampled.shape[1] != 3 or sampled.shape[3] != T * T:
        raise RuntimeError("tex_expand: expected [B, 3, NF, T*T]")
    return _TexExpandFn.apply(sampled, int(T))


def rasterize_fim_wim(faces, image_size: int, near: float = 0.1, far: float = 100.0):
    _chk(faces, "faces")
    B, NF = faces.shape[0], faces.shape[1]
    L = lib()
    ws = torch.empty(int(L.jaf_rasterize_workspace(B, NF, image_size)), device=faces.device, dtype=torch.uint8)
    fim = torch.empty((B, image_size, image_size), device=faces.device, dtype=torch.int32)
    wim = torch.empty((B, image_size, image_size, 3), device=faces.device, dtype=torch.float32)
    check(L.jaf_rasterize_fim_wim(_s(), _p(faces), _p(fim), _p(wim), _p(ws), B, NF, image_size, near, far),
          "jaf_rasterize_fim_wim")
    return fim, wim


def flow_warp(src, src_faces, fim, wim, mask=None, align_corners: bool = False):
    """grid_sample(src, cal_bc_transform(src_faces, fim, wim), border) [* mask] in one kernel, forward only
    (src/cal_flow.py:28-39, src/flow_net.py:91); bit-identical to bc_transform -> grid_sample -> mul_bcast."""
    _chk(src, "flow_warp src"); _chk(src_faces, "src_faces"); _chk(fim, "fim", torch.int32); _chk(wim, "wim")
    if mask is not None:
        _chk(mask, "mask")
    B, C, H, W = src.shape
    S, NF = fim.shape[1], src_faces.shape[1]
    out = torch.empty((B, C, S, S), device=src.device, dtype=torch.float32)
    with _hbm("flow_warp_fwd_kernel", 4.0 * B * (C * H * W + 4 * S * S + C * S * S + (mask.shape[1] * S * S if mask is not None else 0))):
        check(lib().jaf_flow_warp_fwd(_s(), _p(src), _p(src_faces), _p(fim), _p(wim), _p(mask), _p(out), B, C, H, W, NF, S,
                                      mask.shape[1] if mask is not None else 1, 1 if align_corners else 0), "jaf_flow_warp_fwd")
    return out


class _BcTransformFn(Function):
    @staticmethod
    def forward(ctx, src_faces, fim, wim):
        B, NF = src_faces.shape[0], src_faces.shape[1]
        S = fim.shape[1]
        T = torch.empty((B, S, S, 2), device=fim.device, dtype=torch.float32)
        check(lib().jaf_bc_transform(_s(), _p(src_faces), _p(fim), _p(wim), _p(T), B, NF, S), "jaf_bc_transform")
        ctx.save_for_backward(fim, wim)
        ctx.shape = (B, NF, S)
        return T

    @staticmethod
    def backward(ctx, dT):
        fim, wim = ctx.saved_tensors
        B, NF, S = ctx.shape
        d = torch.zeros((B, NF, 3, 3), device=dT.device, dtype=torch.float32)
        check(lib().jaf_bc_transform_bwd(_s(), _p(_c(dT)), _p(fim), _p(wim), _p(d), B, NF, S), "jaf_bc_transform_bwd")
        return d, None, None


def bc_transform(src_faces, fim, wim):
    """cal_bc_transform (src/nmr.py:617-659) with the y re-flip of src/cal_flow.py:30-31; differentiable w.r.t. the
    source faces (the index / weight maps carry no gradient, as in the reference)."""
    _chk(src_faces, "src_faces"); _chk(fim, "fim", torch.int32); _chk(wim, "wim")
    return _BcTransformFn.apply(src_faces, fim, wim)
