"""Device-side input pipeline.

`Fusion_dataset_smpl_interval.__getitem__` (src/data.py:640-773) decodes the frames of a sample with cv2, then
normalises every image in float64 NumPy, builds the TransferTexture silhouettes per sample in Python and returns
float tensors; train/4.convLSTM_flowpro_interval.py:216-237 permutes them HWC -> CHW and uploads ~12 float tensors per
step.  Here the decoded uint8 frames are uploaded as they are and `stage4_batch_from_uint8` produces the tensors
`step.Stage4Trainer.train_step` consumes with three kernels (include/jafpro_hip.h, "Device-side input pipeline").
File decoding (cv2.imread / pickle) stays with the caller: SURVEY marks dataset I/O out of scope.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def _u8(t: torch.Tensor, name: str) -> torch.Tensor:
    return ops._chk(t, name, torch.uint8)


def normalize_images(x: torch.Tensor, mode: str = "signed") -> torch.Tensor:
    """uint8 [..., H, W, C] (C = 3) or [..., H, W] -> fp32 [..., C, H, W] / [..., H, W].
    mode "signed": (x/255 - 0.5)*2 (src/data.py:746-750); "unit": x/255 (:739, :751)."""
    _u8(x, "images")
    m = {"signed": 0, "unit": 1}[mode]
    if x.dim() >= 3 and x.shape[-1] == 3:
        lead, (H, W) = x.shape[:-3], x.shape[-3:-1]
        N = int(np.prod(lead)) if lead else 1
        out = torch.empty(tuple(lead) + (3, H, W), device=x.device, dtype=torch.float32)
        check(lib().jaf_u8_hwc_to_f32_chw(ops._s(), ops._p(x), ops._p(out), N, H * W, 3, m), "jaf_u8_hwc_to_f32_chw")
        return out
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib().jaf_u8_hwc_to_f32_chw(ops._s(), ops._p(x), ops._p(out), 1, x.numel(), 1, m), "jaf_u8_hwc_to_f32_chw")
    return out


def iuv_part_mask(iuv255: torch.Tensor) -> torch.Tensor:
    """uint8 [N,S,S,3] -> fp32 [N,3,S,S]: TransferTexture(ones, IUV) (src/data.py:690-695)."""
    _u8(iuv255, "iuv")
    N, S = iuv255.shape[0], iuv255.shape[1]
    out = torch.empty((N, 3, S, S), device=iuv255.device, dtype=torch.float32)
    check(lib().jaf_iuv_part_mask(ops._s(), ops._p(iuv255), ops._p(out), N, S), "jaf_iuv_part_mask")
    return out


def transfer_texture(texture_im: torch.Tensor, iuv255: torch.Tensor, im: Optional[torch.Tensor] = None) -> torch.Tensor:
    """TransferTexture (src/utils.py:369-394), batched, uint8: texture_im [N,800,1200,3] or [800,1200,3] (shared),
    iuv255 [N,S,S,3], optional background im [N,S,S,3] -> [N,S,S,3]."""
    _u8(texture_im, "texture"); _u8(iuv255, "iuv")
    if im is not None:
        _u8(im, "im")
    batched = texture_im.dim() == 4
    AH, AW = texture_im.shape[-3], texture_im.shape[-2]
    N, S = iuv255.shape[0], iuv255.shape[1]
    if batched and texture_im.shape[0] != N:
        raise RuntimeError("transfer_texture: %d atlases for %d IUV maps" % (texture_im.shape[0], N))
    out = torch.empty_like(iuv255)
    check(lib().jaf_transfer_texture_u8(ops._s(), ops._p(texture_im), ops._p(iuv255), ops._p(im), ops._p(out), N, S, AH, AW,
                                        1 if batched else 0), "jaf_transfer_texture_u8")
    return out


def face_bbox_from_iuv(tgt_iuv255: np.ndarray) -> np.ndarray:
    """Host integers (x0, x1, y0, y1) per sample from the target IUV (parts 23/24 = face), src/data.py:699-716.  The
    reference stores them in a uint8 array (:701), so a right/bottom edge of 256 wraps to 0 under the NumPy it pins
    (1.17); a sample without face pixels gets the all-zero box (x0 == x1 == invalid, train/4...py:340)."""
    iuv = np.asarray(tgt_iuv255)
    out = np.zeros((iuv.shape[0], 4), np.int64)
    for i in range(iuv.shape[0]):
        I = iuv[i, :, :, 0]
        Y, X = np.where((I == 23) | (I == 24))
        if X.size == 0:
            continue
        box = [max(int(X.min()) - 2, 0), min(int(X.max()) + 3, 256), max(int(Y.min()) - 2, 0), min(int(Y.max()) + 3, 256)]
        out[i] = [v & 0xff for v in box]
    return out


def stage4_batch_from_uint8(raw: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """raw (device uint8 unless noted): src_texture_u8 [B,T,800,1200,3], src_mask_u8 [B,T,800,1200] (0/255),
    src_img_u8 [B,T,S,S,3], src_IUV0_u8 [B,S,S,3] (IUV of reference 0), tgt_img_u8 [B,S,S,3], tgt_IUV_u8 [B,S,S,3],
    smpl_real_mask_u8 [B,S,S,3]; fp32: bg_noise [B,3,S,S], *_verts*, *_cam*; host: face_bbox (or None: derived from a
    host copy `tgt_IUV_host`).  Returns the batch dict of synth.stage4_batch / Stage4Trainer.train_step."""
    b: Dict[str, torch.Tensor] = {}
    b["src_texture_im"] = normalize_images(raw["src_texture_u8"])                 # :746 + train/4...py:219
    b["src_mask_im"] = normalize_images(raw["src_mask_u8"], "unit")               # :751
    b["src_img"] = normalize_images(raw["src_img_u8"])                            # :749
    b["tgt_img"] = normalize_images(raw["tgt_img_u8"])                            # :750
    b["tgt_IUV"] = normalize_images(raw["tgt_IUV_u8"])                            # :748
    b["tgt_IUV255"] = raw["tgt_IUV_u8"]                                           # :743-744 (data_255)
    b["src_mask_in_image0"] = iuv_part_mask(raw["src_IUV0_u8"])                   # :692-695, used at train/4...py:230
    b["smpl_real_mask"] = normalize_images(raw["smpl_real_mask_u8"], "unit")      # :738-739
    for k in ("bg_noise", "tgt_verts", "src_verts", "tgt_cam", "src_cam", "src_verts_refs", "src_cam_refs"):
        if k in raw:
            b[k] = raw[k]
    if raw.get("face_bbox") is not None:
        b["face_bbox"] = np.asarray(raw["face_bbox"])
    else:
        b["face_bbox"] = face_bbox_from_iuv(raw["tgt_IUV_host"])
    return b
