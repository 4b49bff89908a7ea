"""Stage-1..3 harnesses and checkpoint I/O on the HIP modules.

The reference trains in four stages (README.md:121-124) with one script each; stage 4 lives in
jafpro_amd/step.py, the three earlier ones are strict subsets of it and are restated here on the same
grouped device tensors:

  stage 1  train/1.text_accu_LSTM.py:140-176        Accumulate_LSTM: atlas + masked L1 over three targets,
                                                     Adam 1e-4 with MultiStepLR([100000, 150000], 0.3) stepped per batch
  stage 2  train/2.text_inpaint_convLSTM.py:118-221  accumulate -> common-area mask -> inpaint, per-part masked L1
                                                     over two targets, Adam 1e-4 on both networks
  stage 3  train/3.inpaint_global_convLSTM_FGAN.py:193-400  stage 4 without the flow / propagation: trainable
                                                     background CRN, 3x accumulating face-D AND image-D updates, the face GAN
                                                     term on the NON-detached crop, Adam 1e-4 (G) / 3e-6 (D, face-D)

Checkpoints are `torch.save(module.state_dict())` files named as the scripts name them
(train/3...py:481-494, train/4...py:518-533; README.md:69-73): <prefix>_iter_<count>.pth with prefixes
Accu, inpaint, bg, refine, D, FD, pro; stage 1 writes iter_<count>.pth (train/1...py:266-269).  The mirrors'
state_dicts carry the reference's keys, so files go both ways (tests/golden/checkpoint_pin.json).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .crn_model import CRN_smaller
from .networks import (Accumulate_LSTM, Accumulate_LSTM_no_loss, FaceDiscriminator, ImageDiscriminator, UNet_inpainter,
                       VGG_l1_loss)
from .step import FlatParams, _used_flags, count_faces, face_crops, flush_bn_counters

CKPT_PREFIX = {"accu": "Accu", "inpaint": "inpaint", "bg": "bg", "refine": "refine", "D": "D", "face": "FD", "flow": "pro"}


# ------------------------------------------------------------------------------------------------
# checkpoint I/O
# ------------------------------------------------------------------------------------------------
def checkpoint_path(ckpt_dir: str, name: str, count: int, stage1: bool = False) -> str:
    fn = "iter_%d.pth" % count
    return os.path.join(ckpt_dir, fn if stage1 else "%s_%s" % (CKPT_PREFIX[name], fn))


def save_checkpoints(ckpt_dir: str, count: int, modules: Dict[str, nn.Module], stage1: bool = False) -> Dict[str, str]:
    """One file per module, CPU tensors under the reference's state_dict keys (what `Module.module.state_dict()`
    of the DataParallel-wrapped reference holds)."""
    os.makedirs(ckpt_dir, exist_ok=True)
    out = {}
    for name, m in modules.items():
        path = checkpoint_path(ckpt_dir, name, count, stage1)
        torch.save({k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, path)
        out[name] = path
    return out


def load_checkpoint(module: nn.Module, path: str, strict: bool = True):
    """`module.load_state_dict(torch.load(path))` (train/4...py:121-140) for a mirror living on the GPU: values are
    copied INTO the existing parameter storage (a trainer's flat buffers stay intact) and every cached packed-weight image
    is dropped."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    res = module.load_state_dict(sd, strict=strict)
    ops.invalidate_packed_weights()
    return res


# ------------------------------------------------------------------------------------------------
# shared pieces
# ------------------------------------------------------------------------------------------------
def _subset(t: torch.Tensor, used: Sequence[int]) -> torch.Tensor:
    used = list(used)
    return t.contiguous() if used == list(range(t.shape[1])) else t[:, used].contiguous()


def _one_hot(n: int, i: int, device) -> torch.Tensor:
    f = torch.zeros(n, dtype=torch.int32)
    f[i] = 1
    return f.to(device)


def multistep_lr(base: float, step: int, milestones=(100000, 150000), gamma: float = 0.3) -> float:
    """lr of optimiser step number `step` (1-based) under MultiStepLR stepped once per batch (train/1...py:90,174-175)."""
    return base * gamma ** sum(1 for m in milestones if step - 1 >= m)


# ------------------------------------------------------------------------------------------------
# stage 1
# ------------------------------------------------------------------------------------------------
class Stage1Trainer:
    def __init__(self, model: Optional[Accumulate_LSTM] = None, lr: float = 1e-4):
        self.model = model if model is not None else Accumulate_LSTM().cuda()
        self.model.train()
        self.flat = {"accu": FlatParams(self.model)}
        self.lr = lr

    def train_step(self, batch: Dict[str, torch.Tensor], used: Sequence[int] = (0, 1, 2, 3)) -> Dict[str, torch.Tensor]:
        """batch: src_texture_im [B,4,3,800,1200], src_mask_im [B,4,800,1200] {0,1}, tgt_texture_im [B,3,3,800,1200],
        tgt_mask_im [B,3,800,1200] {0,1}; `used` = the random reference subset (train/1...py:140-149), whose complement's
        masks are zeroed (:162-165)."""
        f = self.flat["accu"]
        f.zero_grad()
        tex = _subset(batch["src_texture_im"], used)
        src_mask = batch["src_mask_im"]
        keep = torch.zeros(src_mask.shape[1], device=src_mask.device, dtype=src_mask.dtype)
        keep[list(used)] = 1
        src_mask = (src_mask * keep.view(1, -1, 1, 1)).to(torch.uint8)
        tgt_mask = batch["tgt_mask_im"].to(torch.uint8)
        atlas, loss = self.model.forward_atlas(tex, src_mask.unsqueeze(2), tgt_mask.unsqueeze(2), batch["tgt_texture_im"])
        loss.backward()
        f.adam(multistep_lr(self.lr, f.step_count + 1))
        return {"total_loss": loss.detach(), "output_texture": atlas.detach()}

    def save(self, ckpt_dir: str, count: int):
        return save_checkpoints(ckpt_dir, count, {"accu": self.model}, stage1=True)


# ------------------------------------------------------------------------------------------------
# stage 2
# ------------------------------------------------------------------------------------------------
def texture_forward(accu: Accumulate_LSTM_no_loss, inpaint: UNet_inpainter, batch, used):
    """accumulate -> common-area mask -> inpaint on grouped tensors (train/2...py:160-196 == train/4...py:269-300)."""
    T_all = batch["src_texture_im"].shape[1]
    x = ops.atlas_to_parts(_subset(batch["src_texture_im"], used))
    a = accu.forward_grouped(x, len(list(used)))
    masked = ops.part_mask_mul(a, batch["src_mask_im"].float().contiguous(), _used_flags(T_all, used, a.device))
    return a, masked, inpaint.forward_grouped(masked)


class Stage2Trainer:
    def __init__(self, accu: Accumulate_LSTM_no_loss, inpaint: UNet_inpainter, lr: float = 1e-4, num_target: int = 2):
        self.accu, self.inpaint = accu, inpaint
        accu.train(); inpaint.train()
        self.flat = {"accu": FlatParams(accu), "inpaint": FlatParams(inpaint)}
        self.lr, self.num_target = lr, num_target

    def train_step(self, batch, used=(0, 1, 2, 3)):
        """loss = sum over the targets z and the 24 parts of L1mean(inpaint_p * m_z,p , tgt_z,p * m_z,p)
        (train/2...py:197-217).  Every part has the same element count, so the 24 per-part means of one target are
        24 x the mean over the whole masked atlas: one reduction per target instead of 24."""
        for f in self.flat.values():
            f.zero_grad()
        _, _, inp = texture_forward(self.accu, self.inpaint, batch, used)
        tm = batch["tgt_mask_im"].float().contiguous()
        Z = tm.shape[1]
        tgt_parts = ops.atlas_to_parts(batch["tgt_texture_im"][:, :self.num_target].contiguous())    # image z*B + b
        B = inp.shape[0]
        total = None
        for z in range(self.num_target):
            flag = _one_hot(Z, z, inp.device)
            pred = ops.part_mask_mul(inp, tm, flag)
            real = ops.part_mask_mul(tgt_parts[z * B:(z + 1) * B].contiguous(), tm, flag)
            term = ops.l1_loss(pred, real, 24.0)
            total = term if total is None else total + term
        total = total.squeeze(0)
        total.backward()
        for n in ("accu", "inpaint"):
            self.flat[n].adam(self.lr)
        return {"total_loss": total.detach(), "inpaint": inp.detach()}

    def save(self, ckpt_dir: str, count: int):
        """train/2...py:291-296 writes `iter_N.pth` into an accumulate and an inpaint directory; here one directory with the
        stage-3/4 prefixes, which is what stage 3 loads (`accu_iter_20000.pth` / `inpaint_iter_20000.pth`, train/3...py:122-129)."""
        return save_checkpoints(ckpt_dir, count, {"accu": self.accu, "inpaint": self.inpaint})


# ------------------------------------------------------------------------------------------------
# stage 3
# ------------------------------------------------------------------------------------------------
LRS3 = {"accu": 1e-4, "inpaint": 1e-4, "bg": 1e-4, "refine": 1e-4, "D": 3e-6, "face": 3e-6}    # train/3...py:160-165


class Stage3Models(nn.Module):
    def __init__(self, image_size: int = 256):
        super().__init__()
        self.Accu_model = Accumulate_LSTM_no_loss()
        self.inpaint_model = UNet_inpainter()
        self.bg_model = CRN_smaller(3)
        self.refine_model = CRN_smaller(3, fg=True)
        self.discriminator = ImageDiscriminator(ndf=32, input_channel=6)
        self.F_Discriminator = FaceDiscriminator(ndf=32, input_channel=6)
        self.loss_criterion = VGG_l1_loss()
        self.image_size = image_size


def stage3_forward(M: Stage3Models, b, used, align_corners: bool = False):
    """train/3...py:236-284: texture pipeline -> warp -> refine CRN; background CRN WITH gradient (:281-282); fusion."""
    accu, masked, inpaint = texture_forward(M.Accu_model, M.inpaint_model, b, used)
    warp = ops.texture_warp(inpaint, b["tgt_IUV255"], align_corners)
    refine_output, fg_mask = M.refine_model(warp, M.image_size)
    src0 = b["src_img"][:, 0].contiguous()
    bg_mask = 1.0 - b["src_mask_in_image0"]
    bg_incomplete = (bg_mask * src0 + (1.0 - bg_mask) * b["bg_noise"]).contiguous()      # :220-221 (input prep; noise is data)
    bg_output = M.bg_model(bg_incomplete, M.image_size)
    final = ops.blend(refine_output, bg_output, fg_mask)
    return {"final_output": final, "refine_output": refine_output, "fg_mask": fg_mask, "bg_output": bg_output,
            "inpaint_warp": warp, "inpaint": inpaint, "accu": accu, "masked": masked}


class Stage3Trainer:
    def __init__(self, models: Stage3Models, lrs: Optional[Dict[str, float]] = None):
        self.M = models
        models.train()
        self.lrs = dict(LRS3 if lrs is None else lrs)
        self.flat = {"accu": FlatParams(models.Accu_model), "inpaint": FlatParams(models.inpaint_model),
                     "bg": FlatParams(models.bg_model), "refine": FlatParams(models.refine_model),
                     "D": FlatParams(models.discriminator), "face": FlatParams(models.F_Discriminator)}

    def train_step(self, batch, used=(0, 1, 2, 3), align_corners: bool = False):
        prev_ws = ops.set_wgrad_stream(None)          # single stream: this harness is not the benchmarked path
        try:
            return self._train_step(batch, used, align_corners)
        finally:
            ops.set_wgrad_stream(prev_ws)
            flush_bn_counters(self.M)

    def _train_step(self, b, used, align_corners):
        M = self.M
        if count_faces(b["face_bbox"]) == 0:
            raise RuntimeError("no valid face box in the batch (the reference crashes here too, train/3...py:318)")
        for f in self.flat.values():                                             # :193-198
            f.zero_grad()
        g = stage3_forward(M, b, used, align_corners)
        final = g["final_output"]
        target = b["tgt_img"].contiguous()
        loss = M.loss_criterion(final, target)                                   # :285
        face_pred, face_real, face_IUV = face_crops(final, target, b["tgt_IUV"], b["face_bbox"])
        src0 = b["src_img"][:, 0].contiguous()
        face_pred_d, final_d = face_pred.detach(), final.detach()
        for _ in range(3):                                                       # :329-343, no zero_grad in between
            F_errD_real = ops.bce_loss(M.F_Discriminator([face_real, face_IUV]), 1.0)
            F_errD_real.backward()
            F_errD_fake = ops.bce_loss(M.F_Discriminator([face_pred_d, face_IUV]), 0.0)
            F_errD_fake.backward()
            self.flat["face"].adam(self.lrs["face"])
        for _ in range(3):                                                       # :349-364
            errD_real = ops.bce_loss(M.discriminator([target, src0]), 1.0)
            errD_real.backward()
            errD_fake = ops.bce_loss(M.discriminator([final_d, src0]), 0.0)
            errD_fake.backward()
            self.flat["D"].adam(self.lrs["D"])
        errG = ops.bce_loss(M.discriminator([final, src0]), 1.0)                 # :368-370
        F_errG = ops.bce_loss(M.F_Discriminator([face_pred, face_IUV]), 1.0)     # :369,:374 -- NOT detached in stage 3
        total = loss + 2 * errG.squeeze(0) + 2 * F_errG.squeeze(0)               # :377
        total.backward()
        for n in ("accu", "inpaint", "bg", "refine"):                            # :379-382
            self.flat[n].adam(self.lrs[n])
        return {"total_loss": total.detach(), "vgg_l1": loss.detach(), "errD": (errD_real + errD_fake).detach(),
                "errG": errG.detach(), "F_errD": (F_errD_real + F_errD_fake).detach(), "F_errG": F_errG.detach(),
                "final_output": final_d}

    def modules(self) -> Dict[str, nn.Module]:
        M = self.M
        return {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
                "D": M.discriminator, "face": M.F_Discriminator}

    def save(self, ckpt_dir: str, count: int):
        return save_checkpoints(ckpt_dir, count, self.modules())                 # :481-494


def stage4_modules(M) -> Dict[str, nn.Module]:
    """The seven files train/4...py:518-533 writes."""
    return {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
            "D": M.discriminator, "face": M.F_Discriminator, "flow": M.propagater}
