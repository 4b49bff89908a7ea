"""Stage-4 harnesses on the HIP modules: the train step (counterpart of
train/4.convLSTM_flowpro_interval.py:206-413) and the forward-only clip loop (counterpart of
test/conv_pro_test.py:219-279).

The reference scripts cannot run on torch >= 1.7 and do their work on nested Python
lists with a device round trip per sample; this harness keeps the same op sequence and the same
quirks -- three discriminator updates on ACCUMULATING gradients (F10), the face GAN term on a
detached crop (:399), BatchNorm in train mode in the propagater (F9), fresh noise in the
background input (:231, passed in as data) -- on grouped device tensors.
"""
from __future__ import annotations

import collections
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .dist import BackwardOverlap
from .cal_flow import float_estimate
from .crn_model import CRN_smaller
from .flow_net import Propagation3DFlowNet
from .networks import (Accumulate_LSTM_no_loss, FaceDiscriminator, ImageDiscriminator, UNet_inpainter,
                       VGG_l1_loss)

LRS = {"accu": 1e-5, "inpaint": 1e-5, "refine": 1e-5, "flow": 5e-5, "D": 3e-6, "face": 1e-6}   # :169-175


# "on": True while a train step is being captured into a hipGraph; "settling": while GraphedTrainStep warms the host caches
_CAPTURE = {"on": False, "settling": False}
GRAPH_HOT = 2          # a key is captured the GRAPH_HOT-th time it is seen in a row
GRAPH_CACHE = 4      # graphs kept per trainer (each owns a private memory pool)


class FlatParams:
    """All parameters of a module in ONE contiguous buffer (and one gradient buffer): Adam is a
    single launch over it and the RCCL all-reduce a single message per module."""

    def __init__(self, module: nn.Module):
        ps = [p for p in module.parameters() if p.requires_grad]
        sizes = [(p.numel() + 3) // 4 * 4 for p in ps]          # keep every view 16-byte aligned
        total = sum(sizes)
        dev = ps[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        self.m = torch.zeros(total, device=dev, dtype=torch.float32)
        self.v = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p, sz in zip(ps, sizes):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            off += sz
        self.params = ps
        self._offs = None
        self.step_count = 0
        self.dev_state = None           # fp32 [4]: the step count on the device (graph-captured steps, see GraphedTrainStep)
        ops.invalidate_packed_weights()

    def zero_grad(self):
        self.grad.zero_()

    def adam(self, lr: float, done=None):
        # weight gradients may have been written on the side stream: wait for all of it, or -- `done`: an event recorded on that stream
        # behind this module's last weight-gradient launch -- for this module's share only
        # (done=True: the caller has already ordered the current stream behind this module's gradients)
        if done is True:
            pass
        elif done is not None:
            torch.cuda.current_stream().wait_event(done)
        else:
            ops.join_wgrad_stream()
        self.step_count += 1
        ops.adam_step(self.flat, self.grad, self.m, self.v, lr, self.step_count, params=self.params, refresh=True,
                      state=self.dev_state if _CAPTURE["on"] else None)

    def offset(self, i: int) -> int:
        """Element offset of parameter i in the flat buffers."""
        if self._offs is None:
            self._offs = [0]
            for p in self.params:
                self._offs.append(self._offs[-1] + (p.numel() + 3) // 4 * 4)
        return self._offs[i]

    def adam_range(self, lr: float, i0: int, i1: int, done=None, bump: bool = False):
        """Adam over parameters i0 .. i1-1 only (Adam is element-wise: any partition of the buffer takes the same step).  `bump`:
        this call opens the module's optimiser step (the other ranges of the same step pass False).  Not for captured steps."""
        if done is True:
            pass
        elif done is not None:
            torch.cuda.current_stream().wait_event(done)
        else:
            ops.join_wgrad_stream()
        if bump:
            self.step_count += 1
        a, b = self.offset(i0), self.offset(i1)
        ops.adam_step(self.flat[a:b], self.grad[a:b], self.m[a:b], self.v[a:b], lr, self.step_count, params=self.params[i0:i1], refresh=True)

    def sync_dev_state(self):
        """Puts the host's step count on the device (before a capture; the captured launches advance it from there)."""
        if self.dev_state is None:
            self.dev_state = torch.zeros(4, device=self.flat.device, dtype=torch.float32)
        self.dev_state.view(torch.int32)[0] = self.step_count


class Stage4Models(nn.Module):
    def __init__(self, faces: np.ndarray, image_size: int = 256):
        super().__init__()
        self.Accu_model = Accumulate_LSTM_no_loss()
        self.inpaint_model = UNet_inpainter()
        self.bg_model = CRN_smaller(3)
        self.refine_model = CRN_smaller(3, fg=True)
        self.flow_calculator = float_estimate(faces=faces, image_size=image_size)
        self.propagater = Propagation3DFlowNet(9, 32, 2, 3, use_deconv=False)
        self.discriminator = ImageDiscriminator(ndf=32, input_channel=6)
        self.F_Discriminator = FaceDiscriminator(ndf=32, input_channel=6)
        self.loss_criterion = VGG_l1_loss()
        self.image_size = image_size

    def set_train_modes(self):
        """train/4...py:185-191: everything .train() except the frozen background CRN."""
        self.train()
        self.bg_model.eval()
        for p in self.bg_model.parameters():
            p.requires_grad = False


_HOST_KEYS = ("face_bbox", "chosen_frame")      # host integers (src/data.py:702-716): reading them must not sync the device


def _to_dev(batch: Dict[str, np.ndarray], device) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in batch.items():
        if k in _HOST_KEYS:
            out[k] = np.asarray(v)
        else:
            out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device) if isinstance(v, np.ndarray) else v
    return out


_FLAG_CACHE = {}


WGRAD_STREAM = True     # weight gradients on a second side stream
# Where the next clip's preparation (side stream 0) is issued.  "d" (default): its matrix-core part -- the frozen
# background CRN (2.4 TFLOP at B=8) and the frozen VGG's target features -- at the start of the discriminator phase, which is
# ~330 launches of 5-40 us in one dependent chain and leaves most of the chip idle; its renderer part (SMPL projection,
# rasteriser, barycentric flow, flow warp) right before the VGG + GAN loss backward, which it overlaps (north star).
# "bwd": everything right before the loss backward, where the CRN competes with the backward's own kernels.
# Measured at B=8: 68.6 vs 70.4 ms/step.
PREP_AT = ""        # "d" / "b": see Stage4Trainer._prep_at (default: "d" on one GPU, "b" with an active reducer)
# the discriminators' real and generated passes as one batch with per-half BatchNorm statistics (train_step); 0: two passes
D_BATCHED = True          # (tests flip it: the two-call form is the reference of the batched one)
# the VGG + L1 loss and its gradient w.r.t. the generated frame on side stream 3, beside the discriminator phase
VGG_SIDE = True
# the two BCE terms of a batched discriminator pass (and their sum) in one launch each way; 0: two bce_loss calls on slices
BCE_PAIR = True
# optimiser steps of the modules whose backward has finished, issued under the accumulate net's last weight gradients (train_step)
EARLY_ADAM = True
DIST_ISSUE_ON_WGRAD = os.environ.get("JAF_DIST_ISSUE_ON_WGRAD", "1") != "0"     # multi-rank: gradient messages issued from the weight-gradient stream
ATLAS_PACKED = True     # atlas slicing straight into enc1's packed input image
# steps the host may have in flight (0: unbounded).  1.5: the host starts enqueueing step k+1 once the GPU has passed the MIDDLE of step k
# (the mark behind the discriminator updates): one and a half steps' worth of activations alive instead of two
# (round 5, B = 8: 50.95 vs 51.01 ms/step, allocator reserve 43.8 vs 53.1 GB for 20.9 GB of peak allocation: profiles/experiments/round5_run_ahead.txt)
_RA = float(os.environ.get("JAF_RUN_AHEAD", "1.5"))
RUN_AHEAD = int(_RA)
RUN_AHEAD_HALF = _RA == 1.5
RANK_CHECK_EVERY = int(os.environ.get("JAF_RANK_CHECK_EVERY", "200"))     # multi-rank: steps between cross-rank checksum comparisons (0: never)
ACCU_SPLIT = os.environ.get("JAF_ACCU_SPLIT", "1") != "0"     # multi-rank: the accumulate net's gradient message and optimiser step in two parameter ranges


def _side_stream(device=None, which: int = 0) -> "torch.cuda.Stream":
    return ops.aux_stream(which)


def _used_flags(T_all: int, used, device) -> torch.Tensor:
    """int32 [T_all] mask of the reference frames in use, resident on the device (no per-step H2D copy)."""
    k = (T_all, tuple(used), str(device))
    t = _FLAG_CACHE.get(k)
    if t is None:
        f = torch.zeros(T_all, dtype=torch.int32)
        f[list(used)] = 1
        t = f.to(device)
        _FLAG_CACHE[k] = t
    return t


class PreparedClip:
    """Everything of a stage-4 batch that depends on the batch alone (no trainable parameter):
    the frozen background CRN's output, the SMPL projection -> rasterise -> barycentric flow ->
    flow-warp chain (train/4...py:230-231,319-320,325) and, for a train step, the frozen VGG's features
    of the target frame (src/networks.py:118-125).  Produced on the side HIP stream by
    `prepare_clip`, either inside the same step (beside the texture pipeline) or one step ahead
    (beside the previous clip's loss backward, `Stage4Trainer.train_step(next_batch=...)`)."""
    __slots__ = ("batch", "prosrc", "key", "src0", "bg_output", "tsf", "event", "vgg_target", "vgg_event")


_PREP_KEYS = ("src_img", "src_mask_in_image0", "bg_noise", "src_cam", "src_verts", "src_cam_refs", "src_verts_refs",
              "tgt_cam", "tgt_verts", "tgt_img")


def _clip_key(b: Dict[str, torch.Tensor], prosrc: int):
    """Identity of everything `prepare_clip` reads: a loader that recycles the same dict or the same device buffers
    with new contents (copy_ bumps _version) must not be handed the previous clip's preparation."""
    return (prosrc,) + tuple((k, b[k].data_ptr(), b[k]._version, tuple(b[k].shape)) for k in _PREP_KEYS if k in b)


def _source_pose(b: Dict[str, torch.Tensor], prosrc: int):
    """prev_smpl = the SMPL pose of the propagation source, smpl_vertices[:, 1 + random_prosrc] (train/4...py:263-266)."""
    if "src_verts_refs" in b:
        return b["src_cam_refs"][:, prosrc].contiguous(), b["src_verts_refs"][:, prosrc].contiguous()
    if prosrc != 0:
        raise ValueError("prosrc != 0 needs the per-reference SMPL poses (src_verts_refs / src_cam_refs)")
    return b["src_cam"], b["src_verts"]


def prepare_clip(M: Stage4Models, b: Dict[str, torch.Tensor], prosrc: int, with_loss_target: bool = False,
                 part: str = "all", into: Optional[PreparedClip] = None) -> PreparedClip:
    """Issues a clip's parameter-independent preparation on side stream 0.  `part`: "networks" = the frozen background
    CRN and the frozen VGG's target features (matrix-core work), "renderer" = SMPL projection -> rasteriser -> barycentric
    flow -> flow warp (the neural_renderer part the north star names), "all" = both.  The two parts may be issued at
    different points of the previous step (`into` = the PreparedClip the first part returned); `event` always marks the
    end of everything issued so far."""
    main = torch.cuda.current_stream()
    side = _side_stream(main.device)
    side.wait_stream(main)          # the batch tensors were produced on the main stream
    p = into
    if p is None:
        p = PreparedClip()
        p.batch, p.prosrc, p.key = b, prosrc, _clip_key(b, prosrc)
        p.src0 = p.bg_output = p.tsf = p.event = p.vgg_target = p.vgg_event = None
    with torch.cuda.stream(side), torch.no_grad():
        if side != main:
            for k in _PREP_KEYS:    # read on the side stream: a batch freed early must not be reallocated under it
                if k in b:
                    b[k].record_stream(side)
        if part in ("all", "networks"):
            p.src0 = b["src_img"][:, 0].contiguous()
            bg_mask = 1.0 - b["src_mask_in_image0"]                                 # :230-231 (input prep)
            bg_incomplete = (bg_mask * p.src0 + (1.0 - bg_mask) * b["bg_noise"]).contiguous()
            p.bg_output = M.bg_model(bg_incomplete, M.image_size)                   # :319-320
            if with_loss_target and "tgt_img" in b:
                p.vgg_target = M.loss_criterion.target_features(b["tgt_img"])
                p.vgg_event = torch.cuda.Event()
                p.vgg_event.record(side)
        if part in ("all", "renderer"):
            prev_img = b["src_img"][:, prosrc].contiguous()
            src_cam, src_verts = _source_pose(b, prosrc)
            p.tsf = M.flow_calculator(prev_img, [src_cam, None, src_verts, None],
                                      [b["tgt_cam"], None, b["tgt_verts"], None])    # :325
        p.event = torch.cuda.Event()
        p.event.record(side)
    return p


def generator_forward(M: Stage4Models, b: Dict[str, torch.Tensor], used: Sequence[int], prosrc: int,
                      align_corners: bool = False, prepared: Optional[PreparedClip] = None,
                      with_loss_target: bool = False) -> Dict[str, torch.Tensor]:
    """train/4...py:269-331 (== test/conv_pro_test.py:219-279 for one target frame)."""
    B, T_all = b["src_img"].shape[0], b["src_img"].shape[1]
    S = M.image_size
    used = list(used)
    # The frozen background CRN and the SMPL rasterise -> flow -> warp chain depend only on the batch:
    # they run on a side HIP stream beside the texture pipeline (whose deep 13x13 / 25x25 levels
    # launch grids far smaller than the chip) and are joined before the fusion blend.
    main = torch.cuda.current_stream()
    if prepared is None or prepared.key != _clip_key(b, prosrc) or (with_loss_target and "tgt_img" in b and prepared.vgg_target is None):
        prepared = prepare_clip(M, b, prosrc, with_loss_target)
    elif prepared.tsf is None or prepared.bg_output is None:        # only one part was issued ahead: issue the other now
        prepared = prepare_clip(M, b, prosrc, with_loss_target, part="renderer" if prepared.tsf is None else "networks", into=prepared)
    bg_output, tsf = prepared.bg_output, prepared.tsf
    tex = b["src_texture_im"] if len(used) == T_all else b["src_texture_im"][:, used].contiguous()
    # :269-276 -- the 24-part slicing writes the first encoder layer's packed input image itself (no fp32 parts tensor, no packing pass)
    xi = ops.atlas_to_parts_packed(tex.contiguous()) if ATLAS_PACKED else None
    x, ximg = xi if xi is not None else (ops.atlas_to_parts(tex.contiguous()), None)
    accu = M.Accu_model.forward_grouped(x, len(used), x_image=ximg)             # :278
    masked = ops.part_mask_mul(accu, b["src_mask_im"].contiguous(), _used_flags(T_all, used, accu.device))   # :283-298
    inpaint = M.inpaint_model.forward_grouped(masked)                           # :300
    inpaint_warp = ops.texture_warp(inpaint, b["tgt_IUV255"], align_corners)    # :309-312
    refine_output, fg_mask = M.refine_model(inpaint_warp, S)                    # :318
    if prepared.event is not None:          # None: static buffers of a captured step, already ordered before this stream
        main.wait_event(prepared.event)
    for t in (bg_output, tsf, prepared.src0):
        t.record_stream(main)
    fusion = ops.blend(refine_output, bg_output, fg_mask)                       # :321
    pro = M.propagater({"fake_tgt": fusion, "tsf_image": tsf, "use_mask": True,
                        "tgt_smpl_mask": b["smpl_real_mask"], "tgt_IUV": b["tgt_IUV"], "use_IUV": True})
    return {"final_output": pro["pred_target"], "final_mask": pro["weight"], "fusion_output": fusion,
            "refine_output": refine_output, "fg_mask": fg_mask, "bg_output": bg_output, "tsf_image": tsf,
            "inpaint_warp": inpaint_warp, "inpaint": inpaint, "accu": accu, "masked": masked, "prepared": prepared}


def face_crops(final, tgt_img, tgt_IUV, bbox: np.ndarray):
    """train/4...py:338-353.  bbox rows are (x0, x1, y0, y1) host integers (they originate on the
    host, src/data.py:702-716, so no device sync is needed); x0 == x1 marks an invalid face."""
    fp, fr, fi = [], [], []
    for i in range(final.shape[0]):
        x0, x1, y0, y1 = (int(v) for v in bbox[i])
        if x0 == x1:
            continue
        crop = (y0, x0, y1 - y0, x1 - x0)
        fp.append(ops.resize(final[i:i + 1], (64, 64), False, crop=crop))
        fr.append(ops.resize(tgt_img[i:i + 1].contiguous(), (64, 64), False, crop=crop))
        fi.append(ops.resize(tgt_IUV[i:i + 1].contiguous(), (64, 64), False, nearest=True, crop=crop))
    if not fp:
        return None, None, None
    return torch.cat(fp, 0), torch.cat(fr, 0), torch.cat(fi, 0)


def count_faces(bbox: np.ndarray) -> int:
    return int(sum(1 for r in np.asarray(bbox) if int(r[0]) != int(r[1])))


class Stage4Trainer:
    def __init__(self, models: Stage4Models, reducer=None, lrs: Optional[Dict[str, float]] = None):
        self.M = models
        self.M.set_train_modes()
        self.lrs = dict(LRS if lrs is None else lrs)
        self.flat = {
            "accu": FlatParams(models.Accu_model), "inpaint": FlatParams(models.inpaint_model),
            "refine": FlatParams(models.refine_model), "flow": FlatParams(models.propagater),
            "D": FlatParams(models.discriminator), "face": FlatParams(models.F_Discriminator),
        }
        self.reducer = reducer          # jafpro_amd.dist.GradReducer or None (single GPU)
        # where the next clip's preparation is issued: "d" = its networks part (frozen background CRN) beside the discriminator phase and
        # the renderer part beside the generator backward (best on one GPU); "b" = all of it beside the generator backward -- best when
        # RCCL shares the process (two hardware queues, dist.limit_hw_queues): 56.8 -> 55.5 ms with a one-rank group (round4_x4.log)
        self._prep_at = PREP_AT or ("b" if (reducer is not None and reducer.active) else "d")
        self._inflight = collections.deque()      # end-of-step events of the steps the GPU may still be working on (RUN_AHEAD)
        # first parameter of the accumulate net's "complete early" range (see _train_step): the fourth ConvLSTM level's weight
        w4 = getattr(models.Accu_model, "lstm4_w", None)
        self._accu_split = next((i for i, p in enumerate(self.flat["accu"].params) if p is w4), None) if w4 is not None else None
        self._prepared: Optional[PreparedClip] = None
        self.phase_mark = None          # optional callable(name): phase boundaries of train_step (profiling)
        # Rank consistency (the reference's nn.DataParallel re-replicates every module from device 0 in every forward,
        # train/4...py:123-162; here each rank owns a replica for the whole run): parameters, Adam moments, step counts and
        # BatchNorm buffers are taken from rank 0 once, now, and every `check_every` steps the ranks compare an exact checksum of
        # their parameter and moment buffers (BatchNorm running statistics stay rank-local by design: SURVEY 8(e)).
        self.check_every = RANK_CHECK_EVERY
        self._steps_done = 0
        if reducer is not None and reducer.active:
            from .dist import warn_if_hw_queues_unset
            warn_if_hw_queues_unset(reducer)
            self.sync_from_rank0()

    def sync_from_rank0(self) -> None:
        """Every rank takes rank 0's trainable state: flat parameters, Adam moments and step counts of the six modules, and all
        buffers of the model set (BatchNorm running statistics / counters, the frozen networks' weights included)."""
        red = self.reducer
        flush_bn_counters(self.M)
        names = sorted(self.flat)
        red.broadcast([t for n in names for t in (self.flat[n].flat, self.flat[n].m, self.flat[n].v)])
        counts = red.host_broadcast_ints([self.flat[n].step_count for n in names])
        for n, c in zip(names, counts):
            self.flat[n].step_count = c
        frozen = [p.data for p in self.M.parameters() if not p.requires_grad]
        bufs = [b for b in self.M.buffers() if b.is_floating_point() or b.dtype in (torch.int64, torch.int32)]
        red.broadcast(frozen + bufs)
        ops.invalidate_packed_weights()

    def snapshot(self):
        """The training state a step changes -- parameters, Adam moments and step counts of the six modules, every buffer of the
        model set (BatchNorm statistics and counters) -- as device copies; `restore` puts it back."""
        return ({n: (f.flat.clone(), f.m.clone(), f.v.clone(), f.step_count) for n, f in self.flat.items()},
                [(bf, bf.clone()) for bf in self.M.buffers()],
                [(m, m._nbt_pending) for m in self.M.modules() if hasattr(m, "_nbt_pending")])

    def restore(self, snap) -> None:
        ops.join_wgrad_stream()
        for n, (p, m, v, c) in snap[0].items():
            f = self.flat[n]
            f.flat.copy_(p); f.m.copy_(m); f.v.copy_(v)
            f.step_count = c
        for bf, val in snap[1]:
            bf.copy_(val)
        for m, c in snap[2]:
            m._nbt_pending = c
        for f in self.flat.values():                 # the images follow the weights back (per module: the re-pack tables of Adam)
            ops.refresh_packed_weights(f.params)

    def check_rank_consistency(self) -> None:
        """Raises on every rank if any rank's parameters or Adam moments differ in a single bit from rank 0's."""
        red = self.reducer
        if red is None or not red.active:
            return
        ops.join_wgrad_stream()
        names = sorted(self.flat)
        bufs = [t for n in names for t in (self.flat[n].flat, self.flat[n].m, self.flat[n].v)]
        ok = red.consistent(bufs)
        bad = [names[i // 3] + "." + ("param", "adam_m", "adam_v")[i % 3] for i, o in enumerate(ok) if not o]
        if bad:
            raise RuntimeError("data-parallel ranks have diverged after %d steps: %s differ across ranks" % (self._steps_done, ", ".join(bad)))

    def _reduce(self, names: Sequence[str]):
        if self.reducer is not None:
            ops.join_wgrad_stream()
            self.reducer.all_reduce_mean([self.flat[n].grad for n in names])

    def _face_weight(self, n_face: int) -> float:
        """The reference gathers the face crops of the WHOLE batch on device 0 and takes one BCE mean over them
        (train/4...py:338-374), so with N ranks a rank holding n_r of the sum(n) valid faces must weigh its face
        terms by n_r * N / sum(n) for the averaged gradient to be that mean (1.0 whenever the counts are equal).
        The counts are host integers (src/data.py:702-716): they are exchanged on the host (no device sync), and a
        batch without any valid face raises on EVERY rank, as the reference does (:351) -- never a hung collective."""
        if self.reducer is not None and self.reducer.active:
            counts = self.reducer.host_allgather_int(n_face)
            tot = sum(counts)
            if tot == 0:
                raise RuntimeError("no valid face box on any rank (the reference crashes here too, :351)")
            return n_face * len(counts) / tot
        if n_face == 0:
            raise RuntimeError("no valid face box in the batch (the reference crashes here too, :351)")
        return 1.0

    def train_step(self, batch: Dict[str, torch.Tensor], used: Sequence[int] = (0, 1, 2, 3), prosrc: int = 0,
                   align_corners: bool = False, next_batch: Optional[Dict[str, torch.Tensor]] = None,
                   next_prosrc: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """One stage-4 iteration.  `next_batch` (already on the device) is the clip of the NEXT call:
        its parameter-independent preparation (SMPL projection/rasterisation/flow warp, frozen
        background CRN) is issued on the side HIP stream right before this clip's generator loss
        backward and is picked up by that next call (`next_prosrc`: that call's propagation source)."""
        # weight-gradient kernels run on their own stream beside the data gradients (ops.set_wgrad_stream)
        # Bounded run-ahead: the host enqueues a step in about half the time the GPU needs for it, and every step it is ahead keeps
        # that step's activations alive -- blocks that the side streams have touched cannot be handed out again before the GPU has
        # passed them, so the caching allocator answers with fresh hipMalloc segments (0.5-6 GB each, ~27 us per MB of HOST time) step
        # after step: 51 GB reserved after 25 steps at 256 x 256, 113 GB at 512 x 512, and 70-230 ms enqueue spikes that starve the
        # GPU whenever they hit a step without slack (profiles/experiments/round4_x4.log).  Waiting here for the step before the
        # previous one keeps at most RUN_AHEAD steps in flight: the reserve stops growing after the first steps and the host still has
        # a whole step of slack.
        if RUN_AHEAD > 0 and not _CAPTURE["on"] and not _CAPTURE["settling"]:
            if RUN_AHEAD_HALF:
                mid = self.__dict__.pop("_mid_event", None)
                if mid is not None:
                    mid.synchronize()
                self._inflight.clear()
            while len(self._inflight) >= RUN_AHEAD:
                self._inflight.popleft().synchronize()
        prev_ws = ops.set_wgrad_stream(None if not WGRAD_STREAM else ops.aux_stream(1))
        if not _CAPTURE["on"] and not _CAPTURE["settling"]:
            self._last_graph = None      # an eager step in between: a graph's static hand-over slot no longer matches the sequence
        try:
            out = self._train_step(batch, used, prosrc, align_corners, next_batch,
                                   prosrc if next_prosrc is None else next_prosrc)
        finally:
            ops.join_wgrad_stream()
            ops.set_wgrad_stream(prev_ws)
            if RUN_AHEAD > 0 and not _CAPTURE["on"] and not _CAPTURE["settling"]:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._inflight.append(ev)
            if not _CAPTURE["on"]:
                flush_bn_counters(self.M)
        if not _CAPTURE["on"] and not _CAPTURE["settling"]:
            self._steps_done += 1
            if self.check_every > 0 and self.reducer is not None and self.reducer.active and self._steps_done % self.check_every == 0:
                self.check_rank_consistency()
        return out

    def train_step_graphed(self, batch, used: Sequence[int] = (0, 1, 2, 3), prosrc: int = 0, align_corners: bool = False,
                           next_batch=None, next_prosrc: Optional[int] = None):
        """`train_step` through a captured hipGraph (GraphedTrainStep).  A graph is specific to its key -- batch geometry, `used`,
        `prosrc`, the face boxes (host integers that become kernel arguments of the crops), with / without a next clip.  A key
        is captured the SECOND time it is seen in a row (`GRAPH_HOT`): data whose boxes or reference subsets change from clip
        to clip therefore runs the eager step, exactly as `train_step`, and never pays for captures it would not replay; at
        most `GRAPH_CACHE` graphs are kept (least recently used dropped, with their private memory pools).  The call that
        captures runs its own step eagerly; later calls with that key copy the clips into the graph's static buffers and
        replay.  Tensors returned by a replay are the graph's own output buffers: valid until the next replay."""
        nps = prosrc if next_prosrc is None else next_prosrc
        key = (tuple(used), prosrc, nps, bool(align_corners), next_batch is not None, ops.get_precision(),
               tuple(int(v) for v in np.asarray(batch["face_bbox"]).reshape(-1)),
               tuple((k, tuple(v.shape)) for k, v in sorted(batch.items()) if isinstance(v, torch.Tensor)))
        graphs = self.__dict__.setdefault("_graphs", {})
        g = graphs.get(key)
        if g is None:
            seen = self.__dict__.setdefault("_graph_seen", {})
            n = seen.get(key, 0) + 1
            seen.clear()                        # "in a row": another key in between starts the count again
            seen[key] = n
            if n < GRAPH_HOT:
                self._last_graph = None
                return self.train_step(batch, used, prosrc, align_corners, next_batch, nps)
            seen.clear()
            while len(graphs) >= max(1, GRAPH_CACHE):
                graphs.pop(next(iter(graphs)))                      # dicts keep insertion order: the least recently used
            g = graphs[key] = GraphedTrainStep(self, batch, used, prosrc, align_corners, next_batch, nps)
            self._last_graph = g
            return g.first
        graphs[key] = graphs.pop(key)                               # most recently used last
        out = g.step(batch, next_batch, in_sequence=self.__dict__.get("_last_graph") is g)
        self._last_graph = g
        return out

    @staticmethod
    def _generator_backward(total, final, fl, g_vgg):
        """total.backward() (:408).  With the perceptual term taken on the side stream (`fl` given): the adversarial terms are
        differentiated down to the generated frame (which also leaves their never-used deposit in the discriminators' gradient
        buffers, F10), the two frame gradients are added, and the generator is differentiated from there."""
        if fl is None:
            total.backward()
            return
        total.backward()                               # loss is detached: this is 2 errG + 2 F_errG down to `fl` and into D / FD
        ops.axpby(1.0, g_vgg.contiguous(), 1.0, fl.grad)
        final.backward(fl.grad)

    def _train_step(self, batch, used, prosrc, align_corners, next_batch, next_prosrc):
        M, b = self.M, batch
        fw = self._face_weight(count_faces(b["face_bbox"]))      # before anything is enqueued: may raise on every rank
        for f in self.flat.values():                                             # :206-212
            f.zero_grad()
        prepared, self._prepared = self._prepared, None
        g = generator_forward(M, b, used, prosrc, align_corners, prepared, with_loss_target=True)
        final = g["final_output"]
        target = b["tgt_img"].contiguous()
        vgg_target = None
        prep = g["prepared"]
        if prep.vgg_target is not None:         # target features came from the side stream
            if prep.vgg_event is not None:
                torch.cuda.current_stream().wait_event(prep.vgg_event)
            vgg_target = prep.vgg_target
            for t in [vgg_target[0]] + list(vgg_target[1]):
                t.record_stream(torch.cuda.current_stream())
        mark = self.phase_mark or (lambda name: None)
        mark("generator forward")
        # The perceptual loss and ITS gradient w.r.t. the generated frame (VGG forward + data-gradient chain, 2 x 48 GFLOP
        # per sample and nothing else depends on them until the generator backward) run on side stream 3 beside the
        # discriminator phase, whose ~330 launches of 5-40 us leave most of the chip idle; the generator backward then
        # starts from d(loss)/d(final) + d(2 errG)/d(final).  Same arithmetic: the reference's single backward() sums the
        # two contributions at `final` just the same (train/4...py:332,407-408).
        split = VGG_SIDE and torch.is_grad_enabled() and final.requires_grad
        g_vgg = vgg_done = None
        if split:
            main = torch.cuda.current_stream()
            side = ops.aux_stream(3)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fl3 = final.detach().requires_grad_(True)
                loss = M.loss_criterion(fl3, target, target=vgg_target)          # :332
                (g_vgg,) = torch.autograd.grad(loss, fl3)
                loss = loss.detach()
                vgg_done = torch.cuda.Event()
                vgg_done.record(side)
                if side != main:
                    for t in [final, target] + ([vgg_target[0]] + list(vgg_target[1]) if vgg_target is not None else []):
                        t.record_stream(side)
        else:
            loss = M.loss_criterion(final, target, target=vgg_target)            # :332
        mark("VGG+L1 loss forward")
        face_pred, face_real, face_IUV = face_crops(final, target, b["tgt_IUV"], b["face_bbox"])
        src0 = b["src_img"][:, 0].contiguous()
        face_pred_d = face_pred.detach() if face_pred is not None else None
        # The image discriminator's Linear assumes 256x256 inputs (src/networks.py:409).  BASELINE config 5 (512x512, no
        # reference implementation) feeds it 2x average-pooled images; at 256 `dview` is the identity.
        if M.image_size == 512:
            dview = lambda t: ops.avg_pool(t, 2, 2, 0)
        elif M.image_size == 256:
            dview = lambda t: t
        else:
            raise RuntimeError("train_step supports image_size 256 (reference) and 512 (config 5), got %d" % M.image_size)
        target_d, src0_d = dview(target), dview(src0)
        if next_batch is not None and self._prep_at == "d":
            self._prepared = prepare_clip(M, next_batch, next_prosrc, with_loss_target=True, part="networks")
        # ---- face discriminator, one update (:362-374)
        if face_pred is not None and D_BATCHED:
            # real and generated crops through the face discriminator as ONE batch of 2n (see the image discriminator below)
            nfc = face_real.shape[0]
            pf = M.F_Discriminator([torch.cat([face_real, face_pred_d]), torch.cat([face_IUV, face_IUV])], batch_parts=2)
            if BCE_PAIR:        # both terms and their sum in one launch each way, seeded with a cached one (ops.bce_pair)
                F_errD_real, F_errD_fake, fsum = ops.bce_pair(pf, nfc, 1.0, 0.0)
            else:
                F_errD_real = ops.bce_loss(pf[:nfc], 1.0)
                F_errD_fake = ops.bce_loss(pf[nfc:], 0.0)
                fsum = F_errD_real + F_errD_fake
            if fw == 1.0 and BCE_PAIR:
                ops.backward_from(fsum)
            else:
                (fsum if fw == 1.0 else fsum * fw).backward()
        elif face_pred is not None:
            F_errD_real = ops.bce_loss(M.F_Discriminator([face_real, face_IUV]), 1.0)
            (F_errD_real if fw == 1.0 else F_errD_real * fw).backward()
            F_errD_fake = ops.bce_loss(M.F_Discriminator([face_pred_d, face_IUV]), 0.0)
            (F_errD_fake if fw == 1.0 else F_errD_fake * fw).backward()
        else:       # this rank holds no valid face (others do): zero contribution, but it still joins the exchange
            F_errD_real = F_errD_fake = torch.zeros(1, device=final.device)
        self._reduce(["face"])
        self.flat["face"].adam(self.lrs["face"])
        mark("face-D update")
        # ---- image discriminator, three updates on accumulating grads (:380-394, F10)
        final_d = final.detach()
        final_dd = dview(final_d)
        if D_BATCHED:
            # The reference runs D on (target, src) and on (generated, src) in two calls per update and backpropagates the two BCE
            # terms one after the other into the same .grad buffers (:380-394).  Here the two pairs -- the same tensors in all three
            # updates -- form ONE batch of 2B: convolutions and the classifier are per-sample, every BatchNorm takes the statistics
            # of each half separately and updates its running statistics half after half (ops._SplitBatchNormActFn), and the two
            # BCE means are summed before one backward pass: the same numbers with half the launches of a phase that consists
            # of 5-40 us kernels (host 11.3 ms / GPU 6.2 ms for the three updates; profiles/experiments/phases2.py).
            nb_ = target_d.shape[0]
            d_in = [torch.cat([target_d, final_dd]), torch.cat([src0_d, src0_d])]
        for _ in range(3):
            if D_BATCHED:
                pd = M.discriminator(d_in, batch_parts=2)
                if BCE_PAIR:
                    errD_real, errD_fake, dsum = ops.bce_pair(pd, nb_, 1.0, 0.0)
                    ops.backward_from(dsum)
                else:
                    errD_real = ops.bce_loss(pd[:nb_], 1.0)
                    errD_fake = ops.bce_loss(pd[nb_:], 0.0)
                    (errD_real + errD_fake).backward()
            else:
                errD_real = ops.bce_loss(M.discriminator([target_d, src0_d]), 1.0)
                errD_real.backward()
                errD_fake = ops.bce_loss(M.discriminator([final_dd, src0_d]), 0.0)
                errD_fake.backward()
            self._reduce(["D"])
            self.flat["D"].adam(self.lrs["D"])
        mark("D x3 updates")
        if RUN_AHEAD_HALF and not _CAPTURE["on"] and not _CAPTURE["settling"]:
            self._mid_event = torch.cuda.Event()
            self._mid_event.record(torch.cuda.current_stream())
        # ---- generator (:398-413)
        if split:
            fl = final.detach().requires_grad_(True)     # the adversarial term's own leaf: its gradient joins the VGG term's below
            errG = ops.bce_loss(M.discriminator([dview(fl), src0_d]), 1.0)
        else:
            errG = ops.bce_loss(M.discriminator([dview(final), src0_d]), 1.0)
        F_errG = (ops.bce_loss(M.F_Discriminator([face_pred_d, face_IUV]), 1.0) if face_pred is not None
                  else torch.zeros(1, device=final.device))
        if split:
            torch.cuda.current_stream().wait_event(vgg_done)
            for t in (loss, g_vgg):
                t.record_stream(torch.cuda.current_stream())
        total = loss + 2 * errG.squeeze(0) + 2 * F_errG.squeeze(0)
        if next_batch is not None:      # overlaps with the VGG + GAN loss backward below
            self._prepared = prepare_clip(M, next_batch, next_prosrc, with_loss_target=True,
                                          part="renderer" if self._prep_at == "d" else "all", into=self._prepared if self._prep_at == "d" else None)
        if self.reducer is not None and self.reducer.active:
            # each module's gradient messages leave as soon as the backward pass has passed the module's
            # input (reverse graph order), beside the differentiation of the modules upstream of it
            # The messages are issued FROM the weight-gradient stream (RCCL's stream waits for that module's weight gradients, the
            # dependent chain does not wait), and every module takes its optimiser step as soon as its own means have arrived, beside
            # the messages that still travel (JAF_DIST_ISSUE_ON_WGRAD=0: the chain joins the weight-gradient stream before every
            # module's messages and all four optimiser steps follow the last message).
            if DIST_ISSUE_ON_WGRAD:
                ov = BackwardOverlap(self.reducer, issue_stream=ops.wgrad_stream_after_current, before_begin=ops.join_wgrad_stream)
            else:
                ov = BackwardOverlap(self.reducer, before_begin=ops.join_wgrad_stream)
            ov.watch(g["fusion_output"], "flow", [self.flat["flow"].grad])
            ov.watch(g["inpaint_warp"], "refine", [self.flat["refine"].grad])
            ov.watch(g["masked"], "inpaint", [self.flat["inpaint"].grad])
            # The accumulate net's message in two parameter ranges: everything from the fourth ConvLSTM level on in parameter order
            # (levels 4-5 and the decoder: 77 % of its 28 M parameters = 88 of 114 MB) is complete when the backward pass leaves the
            # fourth level -- ops.watch_wgrads reports the moment -- and travels under the rest of the backward pass (the 200 x 200 and
            # 100 x 100 levels, enc1..9); only the remaining 26 MB leave behind the end of the backward pass, where nothing hides them.
            fa = self.flat["accu"]
            k = self._accu_split if (DIST_ISSUE_ON_WGRAD and ACCU_SPLIT and ops.wgrad_stream() is not None) else None
            if k is not None:
                o = fa.offset(k)
                ops.watch_wgrads([p for p in fa.params[k:] if p.dim() == 4], lambda: ov.begin_now("accu_hi", [fa.grad[o:]]))
            self._generator_backward(total, final, fl if split else None, g_vgg)
            ops.watch_wgrads(None)
            if "accu_hi" in ov.fired:
                rest = [(n, [self.flat[n].grad]) for n in ("flow", "refine", "inpaint")] + [("accu_lo", [fa.grad[:o]])]
            else:
                rest = [(n, [self.flat[n].grad]) for n in ("flow", "refine", "inpaint", "accu")]
            if DIST_ISSUE_ON_WGRAD:
                # a module's messages were issued behind its weight gradients, and the chain has waited for the messages: no join needed
                def _step(n):
                    if n == "accu_hi":
                        fa.adam_range(self.lrs["accu"], k, len(fa.params), done=True, bump=True)
                    elif n == "accu_lo":
                        fa.adam_range(self.lrs["accu"], 0, k, done=True)
                    else:
                        self.flat[n].adam(self.lrs[n], done=True)
                ov.finish(rest, each=_step)
                self.overlap_order = list(ov.fired)
                mark("generator loss backward")
                mark("generator Adam x4")
                return {"total_loss": total.detach(), "vgg_l1": loss.detach(), "errD": (errD_real + errD_fake).detach(),
                        "errG": errG.detach(), "F_errD": (F_errD_real + F_errD_fake).detach(), "F_errG": F_errG.detach(),
                        "final_output": final_d}
            ov.finish(rest)
            self.overlap_order = list(ov.fired)
        else:
            # Per-module completion marks on the weight-gradient stream: a module is done when the backward pass has produced the
            # gradient of its INPUT (the hook fires behind the module's last backward node, whose weight gradient is already enqueued).
            # The modules finish in the order refine, inpaint, accumulate; only the accumulate net's last weight gradients (enc_0's
            # 5 x 5 runs alone at the very end) are still in flight when the data-gradient chain ends, so the other optimiser steps
            # and their weight re-packing run under that tail instead of behind it.
            marks_ = {}
            hooks = []
            wst = ops.aux_stream(1) if (WGRAD_STREAM and EARLY_ADAM) else None
            if wst is not None:
                def _mark(name):
                    def hook(_g):
                        ev = torch.cuda.Event()
                        ev.record(wst)
                        marks_[name] = ev
                        return None
                    return hook
                for t, name in ((g["fusion_output"], "flow"), (g["inpaint_warp"], "refine"), (g["masked"], "inpaint")):
                    if t is not None and t.requires_grad:
                        hooks.append(t.register_hook(_mark(name)))
            self._generator_backward(total, final, fl if split else None, g_vgg)
            for h in hooks:
                h.remove()
            early = [n for n in ("refine", "inpaint", "flow") if n in marks_]
            for n in early:
                self.flat[n].adam(self.lrs[n], done=marks_[n])
            mark("generator loss backward")
            for n in ("accu", "inpaint", "refine", "flow"):
                if n not in early:
                    self.flat[n].adam(self.lrs[n])
            mark("generator Adam x4")
            return {"total_loss": total.detach(), "vgg_l1": loss.detach(), "errD": (errD_real + errD_fake).detach(),
                    "errG": errG.detach(), "F_errD": (F_errD_real + F_errD_fake).detach(), "F_errG": F_errG.detach(),
                    "final_output": final_d}
        mark("generator loss backward")
        for n in ("accu", "inpaint", "refine", "flow"):
            self.flat[n].adam(self.lrs[n])
        mark("generator Adam x4")
        return {"total_loss": total.detach(), "vgg_l1": loss.detach(), "errD": (errD_real + errD_fake).detach(),
                "errG": errG.detach(), "F_errD": (F_errD_real + F_errD_fake).detach(), "F_errG": F_errG.detach(),
                "final_output": final_d}


def flush_bn_counters(module: nn.Module) -> None:
    """Adds the host-side BatchNorm call counts (networks._BN._nbt_pending: 55 layers per stage-4 step) to the
    `num_batches_tracked` buffers as ONE batched add, so that the buffers are current between steps for every reader
    (state_dict() flushes by itself; `module.num_batches_tracked`, buffers(), a broadcast or a copy between modules do not)."""
    bns = module.__dict__.get("_bn_list")
    if bns is None:
        bns = module.__dict__["_bn_list"] = [m for m in module.modules() if hasattr(m, "_nbt_pending")]
    pend = [m for m in bns if m._nbt_pending]
    if not pend:
        return
    torch._foreach_add_([m._buffers["num_batches_tracked"] for m in pend], [int(m._nbt_pending) for m in pend])
    for m in pend:
        m._nbt_pending = 0


class _ClipToken:
    """Identity of a clip as the caller holds it.  The token keeps STRONG references to the clip's tensors: while it lives, the
    caching allocator cannot hand their storage to another clip, so "same tensor objects at the same versions" means "same data"
    (a token of addresses alone matched a NEW clip allocated into the freed blocks of a dropped one, both at version 0, and the
    replay then trained on the previously staged clip: ADVICE r4).  Host arrays are compared by value."""
    __slots__ = ("items",)

    def __init__(self, b):
        self.items = tuple((k, v, v._version) if isinstance(v, torch.Tensor) else (k, np.asarray(v).tobytes(), None)
                           for k, v in sorted(b.items()))

    def matches(self, b) -> bool:
        if b is None or len(b) != len(self.items):
            return False
        for k, ref, ver in self.items:
            v = b.get(k)
            if isinstance(ref, torch.Tensor):
                if v is not ref or v._version != ver:
                    return False
            elif isinstance(v, torch.Tensor) or np.asarray(v).tobytes() != ref:
                return False
        return True


def _clip_token(b):
    return None if b is None else _ClipToken(b)


class GraphedTrainStep:
    """One full stage-4 train step captured ONCE into a hipGraph and replayed: ~1500 kernel launches on four streams become
    one graph launch, so the step no longer depends on how fast the host can enqueue (profiles/round2_host_enqueue.txt: 35-58
    ms of Python + launch calls per 65 ms step; eight ranks share one host).  Same kernels, same arithmetic, same order.

    What a captured step fixes, and how the moving parts are handled:
      * geometry and host integers -- batch shapes, the reference subset `used`, the propagation source, the face boxes
        (kernel arguments of the crops) -- are the graph's key (`Stage4Trainer.train_step_graphed`: hot keys only, LRU cap);
      * the clips live in static device buffers.  With a next-clip slot there are THREE sets: `cur` (the clip this replay
        trains on), `nxt` (the clip it prepares) and `stage` (where `step` puts the caller's next clip).  The graph begins
        with cur <- nxt, nxt <- stage: call k, given (B_k, B_k+1), trains on B_k -- which call k-1 staged and prepared --
        and prepares B_k+1;
      * a call that does not continue the sequence (another clip than the one staged last time, or an eager step / another
        graph in between) RESYNCS first: its clip goes into `nxt` and is prepared eagerly into the static hand-over slot;
      * Adam's bias correction needs the step count: it lives on the device (jaf_adam_step_dev), advanced inside the graph;
        the host's counters advance by what the capture counted (one per module, three for the image discriminator);
      * the next clip's preparation (side stream 0) is produced INSIDE the graph into graph-owned tensors and copied into
        the static PreparedClip the next replay starts from -- the cross-step overlap of the eager step, kept;
      * every side stream is joined before the capture ends (weight gradients, re-packed weight images, preparation);
      * a capture must not find a host-side cache to fill (uploads are illegal while capturing): on a cold process the
        step is run up to twice more to warm them, with parameters, Adam moments, step counts and BatchNorm buffers
        restored afterwards -- the capturing call applies exactly ONE step's update, like `train_step`.
    Single-rank only: the gradient exchange of N > 1 ranks stays on the eager path."""

    def __init__(self, trainer: "Stage4Trainer", batch, used, prosrc, align_corners, next_batch, next_prosrc):
        if trainer.reducer is not None and trainer.reducer.active:
            raise RuntimeError("GraphedTrainStep: the multi-rank gradient exchange is not captured; use train_step")
        tr = self.trainer = trainer
        self.used, self.prosrc, self.align, self.next_prosrc = tuple(used), prosrc, align_corners, next_prosrc
        clone = lambda b: {k: (v.clone() if isinstance(v, torch.Tensor) else np.array(v, copy=True)) for k, v in b.items()}
        self.cur = clone(batch)
        self.nxt = clone(next_batch) if next_batch is not None else None
        self.stage = clone(next_batch) if next_batch is not None else None
        self.staged_token = _clip_token(next_batch)
        # THIS call's step runs eagerly on the static buffers: it fills the host-side caches (plans, packed weight images,
        # constants, LDS opt-ins) and leaves the preparation of `nxt` behind; the capture below executes nothing
        _CAPTURE["settling"] = True
        try:
            before = ops.cache_census()
            out = tr.train_step(self.cur, self.used, prosrc, align_corners, next_batch=self.nxt, next_prosrc=next_prosrc)
            self.first = {k: v.clone() for k, v in out.items()}
            self.settle_steps = 1
            # a cold process builds its weight images (and the argument tables of their re-packing) over its first two steps
            # -- the discriminator's data-gradient images only exist after the first generator backward.  Warm-up steps leave
            # no trace in the training state.
            while ops.cache_census() != before and self.settle_steps < 3:
                snap = self._snapshot()
                before = ops.cache_census()
                tr._prepared = None
                tr.train_step(self.cur, self.used, prosrc, align_corners, next_batch=self.nxt, next_prosrc=next_prosrc)
                self._restore(snap)
                self.settle_steps += 1
        finally:
            _CAPTURE["settling"] = False
        torch.cuda.synchronize()
        self.prep = tr._prepared                     # PreparedClip in ordinary memory: the graph's static hand-over slot
        if self.prep is not None:
            self.prep.event = self.prep.vgg_event = None
        counts = {n: f.step_count for n, f in tr.flat.items()}
        for f in tr.flat.values():
            f.sync_dev_state()
        ops.reset_pack_events()
        self.graph = torch.cuda.CUDAGraph()
        _CAPTURE["on"] = True
        try:
            with torch.cuda.graph(self.graph):
                main = torch.cuda.current_stream()
                if self.nxt is not None:             # this replay's clip is what the previous one prepared: nxt -> cur, stage -> nxt
                    for k, v in self.cur.items():
                        if isinstance(v, torch.Tensor):
                            v.copy_(self.nxt[k])
                    for k, v in self.nxt.items():
                        if isinstance(v, torch.Tensor):
                            v.copy_(self.stage[k])
                if self.prep is not None:            # (after the copies: they bump the tensors' versions, which the key holds)
                    self.prep.batch, self.prep.key = self.cur, _clip_key(self.cur, prosrc)
                tr._prepared = self.prep
                self.out = tr.train_step(self.cur, self.used, prosrc, align_corners, next_batch=self.nxt, next_prosrc=next_prosrc)
                new = tr._prepared
                if new is not None:                  # hand the next clip's preparation over through the static slot
                    main.wait_event(new.event)
                    if new.vgg_event is not None:
                        main.wait_event(new.vgg_event)
                    for a, b in self._prep_pairs(self.prep, new):
                        a.copy_(b)
                for which in (0, 1, 2, 3):           # nothing may be left running on a side stream when the capture ends
                    st = ops.aux_stream(which)
                    if st != main:
                        main.wait_stream(st)
        finally:
            _CAPTURE["on"] = False
        # the capture advanced the host's counters without running anything: what it counted is what one replay adds
        self.increments = {n: f.step_count - counts[n] for n, f in tr.flat.items()}
        for n, f in tr.flat.items():
            f.step_count = counts[n]
        tr._prepared = None
        ops.reset_pack_events()
        self.pack_entries = ops.pack_cache_entries()         # the graph writes these images: they must outlive it
        self.replays = 0
        self.resyncs = 0

    def _snapshot(self):
        return self.trainer.snapshot()

    def _restore(self, snap):
        self.trainer.restore(snap)

    @staticmethod
    def _prep_pairs(dst: PreparedClip, src: PreparedClip):
        pairs = [(dst.src0, src.src0), (dst.bg_output, src.bg_output), (dst.tsf, src.tsf)]
        if dst.vgg_target is not None:
            pairs.append((dst.vgg_target[0], src.vgg_target[0]))
            pairs += list(zip(dst.vgg_target[1], src.vgg_target[1]))
        return pairs

    @staticmethod
    def _put(dst, src):
        for k, v in dst.items():
            if isinstance(v, torch.Tensor):
                if src[k] is not v and src[k].data_ptr() != v.data_ptr():
                    v.copy_(src[k])
            else:
                dst[k] = np.array(src[k], copy=True)

    def _resync(self, batch):
        """`batch` is not the clip the previous replay staged and prepared (or something else ran in between): put it where
        the graph expects this replay's clip (`nxt`, moved to `cur` by the graph's first nodes) and prepare it eagerly into
        the static hand-over slot."""
        tr = self.trainer
        self._put(self.nxt, batch)
        p = prepare_clip(tr.M, self.nxt, self.prosrc, with_loss_target=True)
        main = torch.cuda.current_stream()
        main.wait_event(p.event)
        if p.vgg_event is not None:
            main.wait_event(p.vgg_event)
        with torch.no_grad():
            for a, b in self._prep_pairs(self.prep, p):
                a.copy_(b)
        self.resyncs += 1

    def step(self, batch, next_batch=None, in_sequence: bool = True):
        """One replay that trains on `batch` and prepares `next_batch` (graphs captured with a next-clip slot)."""
        if self.nxt is None:
            self._put(self.cur, batch)
        else:
            if next_batch is None:
                raise ValueError("this graph was captured with a next clip: pass next_batch")
            if not in_sequence or self.staged_token is None or not self.staged_token.matches(batch):
                self._resync(batch)
            self._put(self.stage, next_batch)
            self.staged_token = _clip_token(next_batch)
        return self.replay()

    def replay(self):
        self.graph.replay()
        self.replays += 1
        for n, f in self.trainer.flat.items():
            f.step_count += self.increments[n]
        return self.out


@torch.no_grad()
def forward_clip(M: Stage4Models, clip: Dict[str, torch.Tensor], used: Sequence[int] = (0, 1, 2, 3),
                 align_corners: bool = False) -> torch.Tensor:
    """Forward-only clip loop, test/conv_pro_test.py:219-279 (BASELINE config 2): accumulate + inpaint +
    background once per clip, then per target frame warp -> refine -> blend -> flow -> propagate.
    clip tensors: src_* as in a stage-4 batch with B clips; per-frame tensors carry a frame axis:
    tgt_IUV255 [B,F,S,S,3], tgt_IUV [B,F,3,S,S], smpl_real_mask [B,F,3,S,S], tgt_verts [B,F,NV,3],
    tgt_cam [B,F,3]; `chosen_frame` [T] host integers = clip positions of the reference frames: frame i
    is propagated from the reference nearest in time, whose SMPL pose is the clip's own pose at that
    position (:256-262, pro_index clipped as there).  The propagater stays in train mode.
    Returns pred_target [B,F,3,S,S]."""
    S = M.image_size
    B, T_all = clip["src_img"].shape[0], clip["src_img"].shape[1]
    used = list(used)
    tex = clip["src_texture_im"] if len(used) == T_all else clip["src_texture_im"][:, used].contiguous()
    accu = M.Accu_model.forward_grouped(ops.atlas_to_parts(tex.contiguous()), len(used))
    masked = ops.part_mask_mul(accu, clip["src_mask_im"].contiguous(), _used_flags(T_all, used, accu.device))
    inpaint = M.inpaint_model.forward_grouped(masked)
    src0 = clip["src_img"][:, 0].contiguous()
    bg_mask = 1.0 - clip["src_mask_in_image0"]
    bg_output = M.bg_model((bg_mask * src0 + (1.0 - bg_mask) * clip["bg_noise"]).contiguous(), S)
    Fn = clip["tgt_IUV255"].shape[1]
    chosen = np.asarray(clip["chosen_frame"]).reshape(-1).astype(np.int64)
    outs = []
    for f in range(Fn):
        src_pro = int(np.argmin(np.abs(f - chosen)))                            # :257-258
        pro_index = int(np.clip(chosen[src_pro], 0, min(30, Fn - 1)))           # :268
        prev_img = clip["src_img"][:, src_pro].contiguous()
        warp = ops.texture_warp(inpaint, clip["tgt_IUV255"][:, f].contiguous(), align_corners)
        refine_output, fg_mask = M.refine_model(warp, S)
        fusion = ops.blend(refine_output, bg_output, fg_mask)
        tsf = M.flow_calculator(prev_img, [clip["tgt_cam"][:, pro_index].contiguous(), None,
                                           clip["tgt_verts"][:, pro_index].contiguous(), None],
                                [clip["tgt_cam"][:, f].contiguous(), None, clip["tgt_verts"][:, f].contiguous(), None])
        pro = M.propagater({"fake_tgt": fusion, "tsf_image": tsf, "use_mask": True,
                            "tgt_smpl_mask": clip["smpl_real_mask"][:, f].contiguous(),
                            "tgt_IUV": clip["tgt_IUV"][:, f].contiguous(), "use_IUV": True})
        outs.append(pro["pred_target"])
    return torch.stack(outs, 1)
