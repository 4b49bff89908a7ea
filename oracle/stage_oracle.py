"""ORACLE -- TEST INFRASTRUCTURE ONLY (tests/ only).

CPU restatements of the reference's stage-1..3 train steps, composed from oracle/torch_oracle.py (each piece pinned
against the reference modules by oracle/make_golden.py) with torch autograd and torch.optim.Adam:
  stage 1  train/1.text_accu_LSTM.py:116-176        (Adam 1e-4, MultiStepLR([100000,150000], 0.3) stepped per batch)
  stage 2  train/2.text_inpaint_convLSTM.py:118-221 (two Adam 1e-4; masked per-part L1 over num_target = 2 targets)
  stage 3  train/3.inpaint_global_convLSTM_FGAN.py:193-382
           (stage 4 without flow / propagation; trainable background CRN; THREE accumulating face-D updates; the face GAN
            generator term on the non-detached crop; Adam 1e-4 x4, 3e-6 x2)
The scripts themselves do not run under torch >= 1.7; the random reference subset and the background noise
are inputs, as in oracle/step_oracle.py.
"""
from __future__ import annotations

from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from . import torch_oracle as O
from .step_oracle import _as_params


def _parts_in(tex, used):
    x_in = []
    for i in range(4):
        for j in range(6):
            x_in.append([tex[:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200] for t in used])
    return x_in


class OracleStage1:
    def __init__(self, sd, lr=1e-4):
        self.sd = _as_params(sd)
        self.opt = torch.optim.Adam([p for p in self.sd.values() if p.requires_grad], lr=lr)
        self.sched = torch.optim.lr_scheduler.MultiStepLR(self.opt, milestones=[100000, 150000], gamma=0.3)

    def train_step(self, b: Dict[str, torch.Tensor], used: Sequence[int] = (0, 1, 2, 3)):
        self.opt.zero_grad(set_to_none=False)
        used = list(used)
        src_mask = b["src_mask_im"].byte().clone()
        for i in range(src_mask.shape[1]):                       # :162-165
            if i not in used:
                src_mask[:, i] = src_mask[:, i] * 0
        src_mask = src_mask.unsqueeze(2).repeat(1, 1, 3, 1, 1)   # :166-167
        tgt_mask = b["tgt_mask_im"].byte().unsqueeze(2).repeat(1, 1, 3, 1, 1)
        atlas, loss = O.accumulate_lstm_loss(self.sd, _parts_in(b["src_texture_im"], used), src_mask, tgt_mask, b["tgt_texture_im"])
        total = loss.sum()
        total.backward()
        self.opt.step()
        self.sched.step()
        return {"total_loss": total.detach(), "output_texture": atlas.detach()}


def texture_forward(sd_accu, sd_inpaint, b, used):
    accu = O.accumulate_forward(sd_accu, _parts_in(b["src_texture_im"], used))
    masked = O.mask_parts(accu, O.common_area_mask(b["src_mask_im"].float(), used))
    return accu, masked, O.inpaint_forward(sd_inpaint, masked)


class OracleStage2:
    def __init__(self, sd_accu, sd_inpaint, lr=1e-4, num_target=2):
        self.sd = {"accu": _as_params(sd_accu), "inpaint": _as_params(sd_inpaint)}
        self.opt = {k: torch.optim.Adam([p for p in v.values() if p.requires_grad], lr=lr) for k, v in self.sd.items()}
        self.num_target = num_target

    def train_step(self, b, used=(0, 1, 2, 3)):
        for o in self.opt.values():
            o.zero_grad(set_to_none=False)
        _, _, inp = texture_forward(self.sd["accu"], self.sd["inpaint"], b, list(used))
        tm = b["tgt_mask_im"].float().unsqueeze(2).repeat(1, 1, 3, 1, 1)          # :167
        total = 0
        for z in range(self.num_target):                                          # :198-217
            for i in range(4):
                for j in range(6):
                    m = tm[:, z, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200]
                    pred = inp[i * 6 + j] * m
                    target = b["tgt_texture_im"][:, z, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200] * m
                    total = total + F.l1_loss(pred, target)
        total.backward()
        for o in self.opt.values():
            o.step()
        return {"total_loss": total.detach(), "inpaint": torch.cat(inp, 1).detach()}


LRS3 = {"accu": 1e-4, "inpaint": 1e-4, "bg": 1e-4, "refine": 1e-4, "D": 3e-6, "face": 3e-6}


class OracleStage3:
    def __init__(self, sds, lrs=None):
        """sds keys: accu, inpaint, bg, refine, D, face, vgg."""
        self.sd = {k: _as_params(v, trainable=k != "vgg") for k, v in sds.items()}
        lrs = dict(LRS3 if lrs is None else lrs)
        self.opt = {k: torch.optim.Adam([p for p in self.sd[k].values() if p.requires_grad], lr=lrs[k]) for k in LRS3}

    def forward(self, b, used, align_corners=False):
        B = b["src_img"].shape[0]
        accu, masked, inpaint = texture_forward(self.sd["accu"], self.sd["inpaint"], b, list(used))
        iuv = b["tgt_IUV255"].numpy()
        warp = torch.stack([O.texture_warp([t[i] for t in inpaint], iuv[i], align_corners) for i in range(B)])
        refine_output, fg_mask = O.crn_smaller_forward(self.sd["refine"], warp, 256, True)
        src0 = b["src_img"][:, 0]
        bg_mask = 1 - b["src_mask_in_image0"]
        bg_output = O.crn_smaller_forward(self.sd["bg"], bg_mask * src0 + (1 - bg_mask) * b["bg_noise"], 256, False)   # with grad (:281-282)
        final = refine_output * fg_mask.repeat(1, 3, 1, 1) + bg_output * (1 - fg_mask.repeat(1, 3, 1, 1))
        return final

    def train_step(self, b, used=(0, 1, 2, 3), align_corners=False):
        for o in self.opt.values():
            o.zero_grad(set_to_none=False)
        final, target = self.forward(b, used, align_corners), b["tgt_img"]
        loss = O.vgg_l1_loss(self.sd["vgg"], final, target)
        fp, fr, fi = [], [], []
        for i in range(final.shape[0]):                                          # :291-306
            x0, x1, y0, y1 = (int(v) for v in b["face_bbox"][i])
            if x0 == x1:
                continue
            fp.append(F.interpolate(final[i:i + 1, :, y0:y1, x0:x1], size=(64, 64), mode="bilinear", align_corners=False))
            fr.append(F.interpolate(target[i:i + 1, :, y0:y1, x0:x1], size=(64, 64), mode="bilinear", align_corners=False))
            fi.append(F.interpolate(b["tgt_IUV"][i:i + 1, :, y0:y1, x0:x1], size=(64, 64), mode="nearest"))
        face_pred, face_real, face_IUV = torch.cat(fp), torch.cat(fr), torch.cat(fi)
        bce = lambda p, t: F.binary_cross_entropy(p, torch.full_like(p, t))
        FD = lambda x: O.discriminator_forward(self.sd["face"], x, True, O.FACE_D_CONVS)
        D = lambda x: O.discriminator_forward(self.sd["D"], x, True, O.IMAGE_D_CONVS)
        src0 = b["src_img"][:, 0]
        for _ in range(3):                                                       # :329-343
            F_errD_real = bce(FD(torch.cat([face_real, face_IUV], 1)), 1.0)
            F_errD_real.backward()
            F_errD_fake = bce(FD(torch.cat([face_pred.detach(), face_IUV], 1)), 0.0)
            F_errD_fake.backward()
            self.opt["face"].step()
        for _ in range(3):                                                       # :349-364
            errD_real = bce(D(torch.cat([target, src0], 1)), 1.0)
            errD_real.backward()
            errD_fake = bce(D(torch.cat([final.detach(), src0], 1)), 0.0)
            errD_fake.backward()
            self.opt["D"].step()
        errG = bce(D(torch.cat([final, src0], 1)), 1.0)
        F_errG = bce(FD(torch.cat([face_pred, face_IUV], 1)), 1.0)               # non-detached crop (:369)
        total = loss.sum() + 2 * errG + 2 * F_errG
        total.backward()
        for k in ("accu", "inpaint", "bg", "refine"):
            self.opt[k].step()
        return {"total_loss": total.detach(), "vgg_l1": loss.detach(), "errD": (errD_real + errD_fake).detach(),
                "errG": errG.detach(), "F_errD": (F_errD_real + F_errD_fake).detach(), "F_errG": F_errG.detach(),
                "final_output": final.detach()}
