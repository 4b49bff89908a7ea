"""ORACLE -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).

CPU restatement of the stage-4 train step, train/4.convLSTM_flowpro_interval.py:206-413, composed
from oracle/torch_oracle.py (each piece pinned bit-exact against the reference modules by
oracle/make_golden.py) with torch autograd and torch.optim.Adam on CPU.  The script itself cannot
run under torch >= 1.7 (int64 torch.full, BCE target shapes), so the sequence is
restated with exactly the F6 patches of SURVEY Appendix C and nothing else:
  * three discriminator updates on gradients that are NOT zeroed in between (:380-394, F10),
  * face GAN generator term on face_pred.detach() (:399),
  * propagater BatchNorm in train mode; background CRN frozen, under no_grad (:319-320),
  * total = loss.sum() + 2*errG + 2*F_errG (:407), Adam lrs of :169-175.
The random reference subset (:249-261) and the fresh background noise (:231) are inputs.

Frame size: the reference is 256 x 256 only (SMPLRenderer(image_size=256), the image discriminator's Linear,
src/networks.py:409).  BASELINE config 5 names 512 x 512 frames, which the reference cannot run; for that geometry this
file applies the SAME restated modules at the frames' own size (every one of them is size-agnostic except that Linear)
and hands the image discriminator 2x average-pooled images.  That extension is this build's definition of config 5, not
the reference's: fixtures made at 512 (tests/golden/step_*512*.npz) are "parity unpinned" beyond the per-module pins.
"""
from __future__ import annotations

from typing import Dict, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import raster_oracle
from . import torch_oracle as O

LRS = {"accu": 1e-5, "inpaint": 1e-5, "refine": 1e-5, "flow": 5e-5, "D": 3e-6, "face": 1e-6}


def _as_params(sd, trainable=True):
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if trainable and t.is_floating_point() and not ("running_" in k):
            t.requires_grad_(True)
        out[k] = t
    return out


class OracleStage4:
    def __init__(self, sds: Dict[str, Dict[str, torch.Tensor]], faces_idx: np.ndarray, lrs=None):
        """sds keys: accu, inpaint, bg, refine, flow, D, face, vgg (state_dicts with reference keys)."""
        self.sd = {k: _as_params(v, trainable=k not in ("bg", "vgg")) for k, v in sds.items()}
        self.faces_idx = faces_idx
        lrs = dict(LRS if lrs is None else lrs)
        self.opt = {k: torch.optim.Adam([p for p in self.sd[k].values() if p.requires_grad], lr=lrs[k])
                    for k in ("accu", "inpaint", "refine", "flow", "D", "face")}

    # train/4...py:269-331
    def generator_forward(self, b: Dict[str, torch.Tensor], used: Sequence[int], prosrc: int, align_corners=False, sd=None):
        sd = self.sd if sd is None else sd           # a rank's view: shared parameters, own BatchNorm buffers
        B, S = b["src_img"].shape[0], b["src_img"].shape[-1]      # S: 256 (reference) or 512 (config 5, see the header)
        used = list(used)
        x_in = []
        for i in range(4):
            for j in range(6):
                x_in.append([b["src_texture_im"][:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200] for t in used])
        accu = O.accumulate_forward(sd["accu"], x_in)
        area = O.common_area_mask(b["src_mask_im"], used)
        masked = O.mask_parts(accu, area)
        inpaint = O.inpaint_forward(sd["inpaint"], masked)
        iuv = b["tgt_IUV255"].numpy()
        warp = torch.stack([O.texture_warp([t[i] for t in inpaint], iuv[i], align_corners) for i in range(B)])
        refine_output, fg_mask = O.crn_smaller_forward(sd["refine"], warp, S, True)
        src0 = b["src_img"][:, 0]
        bg_mask = 1 - b["src_mask_in_image0"]
        bg_incomplete = bg_mask * src0 + (1 - bg_mask) * b["bg_noise"]
        with torch.no_grad():
            bg_output = O.crn_smaller_forward(sd["bg"], bg_incomplete, S, False)
        fusion = refine_output * fg_mask.repeat(1, 3, 1, 1) + bg_output * (1 - fg_mask.repeat(1, 3, 1, 1))
        with torch.no_grad():
            # prev_smpl is the SMPL pose of the chosen propagation source, smpl_vertices[:, 1 + random_prosrc] (:263-266)
            if "src_verts_refs" in b:
                sv, sc = b["src_verts_refs"][:, prosrc], b["src_cam_refs"][:, prosrc]
            else:
                assert prosrc == 0, "prosrc != 0 needs the per-reference poses (src_verts_refs / src_cam_refs)"
                sv, sc = b["src_verts"], b["src_cam"]
            fs = O.project_faces(sv, sc, self.faces_idx)
            ft = O.project_faces(b["tgt_verts"], b["tgt_cam"], self.faces_idx)
            fim, wim = raster_oracle.rasterize_fim_wim(ft.numpy(), S)
            tsf, _ = O.flow_warp(b["src_img"][:, prosrc], fs, torch.from_numpy(fim), torch.from_numpy(wim), align_corners)
        pro = O.propagation_forward(sd["flow"], {"fake_tgt": fusion, "tsf_image": tsf, "use_mask": True,
                                                      "tgt_smpl_mask": b["smpl_real_mask"], "tgt_IUV": b["tgt_IUV"],
                                                      "use_IUV": True}, True)
        return {"final_output": pro["pred_target"], "final_mask": pro["weight"], "fusion_output": fusion,
                "refine_output": refine_output, "fg_mask": fg_mask, "bg_output": bg_output, "tsf_image": tsf,
                "inpaint_warp": warp, "inpaint": torch.cat(inpaint, 1), "accu": torch.cat(accu, 1)}

    # test/conv_pro_test.py:219-279 (BASELINE config 2): texture pipeline and background once per clip, then per
    # target frame warp -> refine -> blend -> flow from the reference nearest in time -> propagate (train-mode BN)
    @torch.no_grad()
    def forward_clip(self, c: Dict[str, torch.Tensor], used=(0, 1, 2, 3), align_corners=False):
        B = c["src_img"].shape[0]
        used = list(used)
        x_in = []
        for i in range(4):
            for j in range(6):
                x_in.append([c["src_texture_im"][:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200] for t in used])
        accu = O.accumulate_forward(self.sd["accu"], x_in)
        masked = O.mask_parts(accu, O.common_area_mask(c["src_mask_im"], used))
        inpaint = O.inpaint_forward(self.sd["inpaint"], masked)
        src0 = c["src_img"][:, 0]
        bg_mask = 1 - c["src_mask_in_image0"]
        bg_output = O.crn_smaller_forward(self.sd["bg"], bg_mask * src0 + (1 - bg_mask) * c["bg_noise"], 256, False)
        Fn = c["tgt_IUV255"].shape[1]
        chosen = np.asarray(c["chosen_frame"]).reshape(-1)
        outs = []
        for f in range(Fn):
            src_pro = int(np.argmin(np.abs(f - chosen)))
            pro_index = int(np.clip(chosen[src_pro], 0, min(30, Fn - 1)))
            iuv = c["tgt_IUV255"][:, f].numpy()
            warp = torch.stack([O.texture_warp([t[i] for t in inpaint], iuv[i], align_corners) for i in range(B)])
            refine_output, fg_mask = O.crn_smaller_forward(self.sd["refine"], warp, 256, True)
            fusion = refine_output * fg_mask.repeat(1, 3, 1, 1) + bg_output * (1 - fg_mask.repeat(1, 3, 1, 1))
            fs = O.project_faces(c["tgt_verts"][:, pro_index], c["tgt_cam"][:, pro_index], self.faces_idx)
            ft = O.project_faces(c["tgt_verts"][:, f], c["tgt_cam"][:, f], self.faces_idx)
            fim, wim = raster_oracle.rasterize_fim_wim(ft.numpy(), 256)
            tsf, _ = O.flow_warp(c["src_img"][:, src_pro], fs, torch.from_numpy(fim), torch.from_numpy(wim), align_corners)
            pro = O.propagation_forward(self.sd["flow"], {"fake_tgt": fusion, "tsf_image": tsf, "use_mask": True,
                                                          "tgt_smpl_mask": c["smpl_real_mask"][:, f], "tgt_IUV": c["tgt_IUV"][:, f],
                                                          "use_IUV": True}, True)
            outs.append(pro["pred_target"])
        return torch.stack(outs, 1)

    def _rank_views(self, n: int):
        """Rank r's view of the state: the SAME parameter tensors (so .grad accumulates the sum over ranks) and,
        for r > 0, its OWN copies of the BatchNorm buffers -- DataParallel replicas / one process per GPU keep
        batch statistics and running stats rank-local (train/4...py:123-162)."""
        if not hasattr(self, "_views"):
            self._views = [self.sd]
        while len(self._views) < n:
            self._views.append({m: {k: (v if v.requires_grad else v.detach().clone()) for k, v in sd.items()}
                                for m, sd in self.sd.items()})
        return self._views[:n]

    def train_step(self, b: Dict[str, torch.Tensor], used=(0, 1, 2, 3), prosrc=0, align_corners=False):
        return self.train_step_ranks([b], used, prosrc, align_corners)[0]

    def train_step_ranks(self, shards, used=(0, 1, 2, 3), prosrc=0, align_corners=False):
        """One step of N data-parallel ranks: the reference step run per shard -- per-shard BatchNorm
        statistics, per-shard batch-mean losses -- with every gradient the MEAN over the shards (each shard's loss is
        scaled by 1/N before backward, gradients accumulate in the shared parameters), then one optimiser step.
        N = 1 is train/4...py:206-413 itself (scale 1.0)."""
        N = len(shards)
        views = self._rank_views(N)
        inv = 1.0 / N
        for o in self.opt.values():
            o.zero_grad(set_to_none=False)
        bce = lambda p, t: F.binary_cross_entropy(p, torch.full_like(p, t))
        R = []
        for b, sd in zip(shards, views):
            g = self.generator_forward(b, used, prosrc, align_corners, sd=sd)
            final, target = g["final_output"], b["tgt_img"]
            loss = O.vgg_l1_loss(sd["vgg"], final, target)
            fp, fr, fi = [], [], []
            for i in range(final.shape[0]):                                         # :338-353
                x0, x1, y0, y1 = (int(v) for v in b["face_bbox"][i])
                if x0 == x1:
                    continue
                fp.append(F.interpolate(final[i:i + 1, :, y0:y1, x0:x1], size=(64, 64), mode="bilinear", align_corners=False))
                fr.append(F.interpolate(target[i:i + 1, :, y0:y1, x0:x1], size=(64, 64), mode="bilinear", align_corners=False))
                fi.append(F.interpolate(b["tgt_IUV"][i:i + 1, :, y0:y1, x0:x1], size=(64, 64), mode="nearest"))
            R.append({"sd": sd, "final": final, "target": target, "loss": loss, "src0": b["src_img"][:, 0], "nf": len(fp),
                      "face_pred": torch.cat(fp) if fp else None, "face_real": torch.cat(fr) if fp else None,
                      "face_IUV": torch.cat(fi) if fp else None})
        # the reference takes ONE mean over the face crops of the whole batch (:338-374): a shard holding n_r of the
        # sum(n) faces weighs its (per-shard mean) face terms by n_r * N / sum(n); 1.0 when the counts are equal
        nf_tot = sum(r["nf"] for r in R)
        assert nf_tot > 0, "no valid face box in the batch (the reference crashes here too, :351)"
        for r in R:
            r["fw"] = r["nf"] * N / nf_tot
        FD = lambda r, x: O.discriminator_forward(r["sd"]["face"], x, True, O.FACE_D_CONVS)
        S = shards[0]["src_img"].shape[-1]
        assert S in (256, 512), S
        dview = (lambda x: F.avg_pool2d(x, 2, 2)) if S == 512 else (lambda x: x)          # header: config 5 only
        D = lambda r, x: O.discriminator_forward(r["sd"]["D"], dview(x), True, O.IMAGE_D_CONVS)
        for r in R:
            if r["nf"] == 0:
                r["F_errD_real"] = r["F_errD_fake"] = torch.zeros(())
                continue
            r["F_errD_real"] = bce(FD(r, torch.cat([r["face_real"], r["face_IUV"]], 1)), 1.0)
            (r["F_errD_real"] * (inv * r["fw"])).backward()
            r["F_errD_fake"] = bce(FD(r, torch.cat([r["face_pred"].detach(), r["face_IUV"]], 1)), 0.0)
            (r["F_errD_fake"] * (inv * r["fw"])).backward()
        self.opt["face"].step()
        for _ in range(3):                                                       # no zero_grad inside (F10)
            for r in R:
                r["errD_real"] = bce(D(r, torch.cat([r["target"], r["src0"]], 1)), 1.0)
                (r["errD_real"] * inv).backward()
                r["errD_fake"] = bce(D(r, torch.cat([r["final"].detach(), r["src0"]], 1)), 0.0)
                (r["errD_fake"] * inv).backward()
            self.opt["D"].step()
        outs = []
        for r in R:
            errG = bce(D(r, torch.cat([r["final"], r["src0"]], 1)), 1.0)
            F_errG = bce(FD(r, torch.cat([r["face_pred"].detach(), r["face_IUV"]], 1)), 1.0) if r["nf"] else torch.zeros(())
            total = r["loss"].sum() + 2 * errG + 2 * F_errG
            (total * inv).backward()
            outs.append({"total_loss": total.detach(), "vgg_l1": r["loss"].detach(),
                         "errD": (r["errD_real"] + r["errD_fake"]).detach(), "errG": errG.detach(),
                         "F_errD": (r["F_errD_real"] + r["F_errD_fake"]).detach(), "F_errG": F_errG.detach(),
                         "final_output": r["final"].detach()})
        for k in ("accu", "inpaint", "refine", "flow"):
            self.opt[k].step()
        return outs
