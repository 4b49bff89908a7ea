"""ORACLE -- TEST INFRASTRUCTURE ONLY.  NumPy restatement of the tensor preparation of
Fusion_dataset_smpl_interval.__getitem__ (src/data.py:640-773), TransferTexture (src/utils.py:369-394) and the
permutes / casts of train/4.convLSTM_flowpro_interval.py:216-237, on already-decoded uint8 frames (cv2.imread is file
I/O, out of scope).  float64 arithmetic and the final .float() cast exactly as the reference.
Pinning (round 3): `transfer_texture` reproduces the reference's own TransferTexture exactly -- oracle/make_golden.py g_data
takes that function out of src/utils.py's syntax tree (the module itself needs tensorflow / cv2 / moviepy) and runs it on
seeded inputs: max |diff| = 0, outputs committed as tests/golden/transfer_texture.npz.  The rest (the normalisations and
permutes of src/data.py:736-763, the face box of :699-716) sits inside a dataset method that needs cv2 and files: pinned by
reading the cited lines only."""
from __future__ import annotations

import numpy as np


def transfer_texture(TextureIm, IUV, im=None):
    """src/utils.py:369-394 verbatim in behaviour (TextureIm (800,1200,3) u8, IUV (S,S,3) u8)."""
    output_img = np.zeros(IUV.shape[:2] + (3,), np.uint8)
    U = np.rint(IUV[:, :, 1] / 255. * 199.).astype(np.uint8)
    V = np.rint(IUV[:, :, 2] / 255. * 199.).astype(np.uint8)
    for partId in range(1, 25):
        i_cor = (partId - 1) // 6
        j_cor = partId - i_cor * 6 - 1
        tex = TextureIm[i_cor * 200:i_cor * 200 + 200, j_cor * 200:j_cor * 200 + 200, :]
        x, y = np.where(IUV[:, :, 0] == partId)
        output_img[x, y, :] = tex[U[x, y], 199 - V[x, y], :]
    if im is not None:
        BG_MASK = output_img == 0
        output_img[BG_MASK] = im[BG_MASK]
    return output_img


def face_bbox(tgt_IUV):
    """src/data.py:699-716 for one target frame, with the uint8 storage of :701 (values wrap mod 256 under numpy 1.17)."""
    Y1, X1 = np.where(tgt_IUV[:, :, 0] == 23)
    Y2, X2 = np.where(tgt_IUV[:, :, 0] == 24)
    X_con, Y_con = np.concatenate([X1, X2]), np.concatenate([Y1, Y2])
    if X_con.size == 0:
        return np.zeros(4, np.int64)
    box = [max(np.min(X_con) - 2, 0), min(np.max(X_con) + 3, 256), max(np.min(Y_con) - 2, 0), min(np.max(Y_con) + 3, 256)]
    return np.array([int(v) % 256 for v in box], np.int64)


def stage4_batch(raw):
    """raw: the uint8 arrays of jafpro_amd.data.stage4_batch_from_uint8 (NumPy) -> float32 batch arrays."""
    norm = lambda x: ((x / 255.0 - 0.5) * 2)
    B = raw["tgt_img_u8"].shape[0]
    b = {}
    b["src_texture_im"] = norm(raw["src_texture_u8"]).transpose(0, 1, 4, 2, 3).astype(np.float32)
    b["src_mask_im"] = (raw["src_mask_u8"] / 255.0).astype(np.float32)
    b["src_img"] = norm(raw["src_img_u8"]).transpose(0, 1, 4, 2, 3).astype(np.float32)
    b["tgt_img"] = norm(raw["tgt_img_u8"]).transpose(0, 3, 1, 2).astype(np.float32)
    b["tgt_IUV"] = norm(raw["tgt_IUV_u8"]).transpose(0, 3, 1, 2).astype(np.float32)
    b["tgt_IUV255"] = raw["tgt_IUV_u8"]
    ones = np.ones((800, 1200, 3), np.uint8)
    b["src_mask_in_image0"] = np.stack([transfer_texture(ones, raw["src_IUV0_u8"][i]) for i in range(B)]).transpose(0, 3, 1, 2).astype(np.float32)
    b["smpl_real_mask"] = (raw["smpl_real_mask_u8"] / 255.0).transpose(0, 3, 1, 2).astype(np.float32)
    b["face_bbox"] = np.stack([face_bbox(raw["tgt_IUV_u8"][i]) for i in range(B)])
    return b
