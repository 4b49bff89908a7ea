"""Portable, seeded synthetic weights and stage-4 inputs.

Everything comes from NumPy's PCG64 (`default_rng`), keyed by name, so the survey container
(where the goldens are made from the reference), the CPU tests and the GPU box regenerate the
same tensors without relying on torch's RNG streams.  No dataset, SMPL asset or checkpoint of
the reference is available offline: the mesh is a procedural closed surface with
the SMPL counts (6890 vertices, 13776 faces).
"""
from __future__ import annotations

import functools
import zlib
from typing import Dict, Mapping

import numpy as np

NV, NF = 6890, 13776


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def synth_state_dict(shapes: Mapping[str, tuple], seed: int) -> Dict[str, np.ndarray]:
    """Deterministic values for every entry of a state_dict, independent of key order."""
    out = {}
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        r = _rng(seed, key)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, np.int64)
        elif leaf == "running_mean":
            out[key] = r.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif leaf == "running_var":
            out[key] = r.uniform(0.8, 1.2, shape).astype(np.float32)
        elif leaf == "gamma":
            out[key] = r.uniform(0.5, 1.0, shape).astype(np.float32)
        elif leaf == "beta":
            out[key] = r.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif len(shape) >= 2:                       # conv / linear weight: He-uniform
            fan_in = int(np.prod(shape[1:]))
            bound = np.sqrt(6.0 / fan_in)
            out[key] = r.uniform(-bound, bound, shape).astype(np.float32)
        elif leaf == "weight":                      # BatchNorm scale
            out[key] = r.uniform(0.5, 1.5, shape).astype(np.float32)
        else:                                       # biases
            out[key] = r.uniform(-0.1, 0.1, shape).astype(np.float32)
    return out


def load_synth(module, seed: int):
    """Fills a torch module (reference or jafpro_amd) in place; returns it."""
    import torch
    sd = module.state_dict()
    vals = synth_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    return module


# ------------------------------------------------------------------------------------------------
# procedural body mesh
# ------------------------------------------------------------------------------------------------
def body_mesh():
    verts, faces = _body_mesh()
    return verts.copy(), faces.copy()


@functools.lru_cache(maxsize=1)
def _body_mesh():
    """Closed genus-0 surface with exactly 6890 vertices / 13776 triangles (82 rings x 84 segments
    + 2 poles), front faces counter-clockwise for the rasteriser's cull test
    (rasterize_cuda_kernel.cu:40 keeps (x1-x0)(y2-y0) - (y1-y0)(x2-x0) >= 0 after the y-flip)."""
    R, S = 82, 84
    verts = np.zeros((NV, 3), np.float64)
    verts[0] = (0.0, 0.85, 0.0)
    verts[NV - 1] = (0.0, -0.85, 0.0)
    for r in range(R):
        th = np.pi * (r + 1) / (R + 1)
        for s in range(S):
            ph = 2 * np.pi * s / S
            rad = np.sin(th) * (1.0 + 0.15 * np.sin(3 * th))
            verts[1 + r * S + s] = (0.34 * rad * np.cos(ph), 0.85 * np.cos(th), 0.22 * rad * np.sin(ph))
    faces = []
    ring = lambda r, s: 1 + r * S + (s % S)
    for s in range(S):
        faces.append((0, ring(0, s + 1), ring(0, s)))
        faces.append((NV - 1, ring(R - 1, s), ring(R - 1, s + 1)))
    for r in range(R - 1):
        for s in range(S):
            a, b, c, d = ring(r, s), ring(r, s + 1), ring(r + 1, s), ring(r + 1, s + 1)
            faces.append((a, b, c))
            faces.append((b, d, c))
    faces = np.asarray(faces, np.int32)
    assert faces.shape == (NF, 3)
    return verts.astype(np.float32), faces


def posed_vertices(seed: int, name: str, batch: int) -> np.ndarray:
    """[batch, 6890, 3]: small rigid rotation about y and z plus per-vertex jitter."""
    base, _ = body_mesh()
    r = _rng(seed, name)
    out = np.zeros((batch, NV, 3), np.float32)
    for b in range(batch):
        ay, az = r.uniform(-0.35, 0.35), r.uniform(-0.15, 0.15)
        cy, sy, cz, sz = np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        v = base.astype(np.float64) @ (Rz @ Ry).T + r.normal(0, 0.002, base.shape)
        out[b] = v.astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------
# stage-4 batch
# ------------------------------------------------------------------------------------------------
def iuv255(seed: int, name: str, batch: int, size: int = 256) -> np.ndarray:
    """uint8 [batch, S, S, 3] (I, U, V): an elliptical body region cut into a 6x4 grid of the 24
    parts (~35 % foreground), U a ramp down each cell, V a ramp across it."""
    r = _rng(seed, name)
    out = np.zeros((batch, size, size, 3), np.uint8)
    yy, xx = np.mgrid[0:size, 0:size]
    for b in range(batch):
        cx, cy = size / 2 + r.uniform(-8, 8), size / 2 + r.uniform(-8, 8)
        rx, ry = size * r.uniform(0.26, 0.30), size * r.uniform(0.40, 0.44)
        inside = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
        gy = np.clip(((yy - (cy - ry)) / (2 * ry) * 6).astype(int), 0, 5)
        gx = np.clip(((xx - (cx - rx)) / (2 * rx) * 4).astype(int), 0, 3)
        part = gy * 4 + gx + 1
        fy = ((yy - (cy - ry)) / (2 * ry) * 6) % 1.0
        fx = ((xx - (cx - rx)) / (2 * rx) * 4) % 1.0
        out[b, ..., 0] = np.where(inside, part, 0)
        out[b, ..., 1] = np.where(inside, np.clip(fy * 255, 0, 255), 0)
        out[b, ..., 2] = np.where(inside, np.clip(fx * 255, 0, 255), 0)
    return out


def rect_masks(seed: int, name: str, shape, n_rect: int = 8) -> np.ndarray:
    """{0,1} float32 masks: per 200x200 atlas cell a union of random rectangles."""
    r = _rng(seed, name)
    *lead, H, W = shape
    m = np.zeros((int(np.prod(lead)) if lead else 1, H, W), np.float32)
    for i in range(m.shape[0]):
        for cy in range(0, H, 200):
            for cx in range(0, W, 200):
                for _ in range(n_rect):
                    y0, x0 = r.integers(0, 160), r.integers(0, 160)
                    h, w = r.integers(20, 90), r.integers(20, 90)
                    m[i, cy + y0:min(cy + y0 + h, cy + 200, H), cx + x0:min(cx + x0 + w, cx + 200, W)] = 1.0
    return m.reshape(shape)


def uniform(seed: int, name: str, shape, lo=-1.0, hi=1.0) -> np.ndarray:
    return _rng(seed, name).uniform(lo, hi, shape).astype(np.float32)


def normal(seed: int, name: str, shape) -> np.ndarray:
    return _rng(seed, name).normal(0.0, 1.0, shape).astype(np.float32)


def stage4_batch(seed: int, B: int, T: int = 4, S: int = 256) -> Dict[str, np.ndarray]:
    """One synthetic stage-4 batch in the layouts of Fusion_dataset_smpl_interval after the
    permutes at train/4.convLSTM_flowpro_interval.py:216-237 (src/data.py:640-773)."""
    d: Dict[str, np.ndarray] = {}
    d["src_img"] = uniform(seed, "src_img", (B, T, 3, S, S))
    d["src_texture_im"] = uniform(seed, "src_texture_im", (B, T, 3, 800, 1200))
    d["src_mask_im"] = rect_masks(seed, "src_mask_im", (B, T, 800, 1200))
    # person silhouette of the first reference in the image plane, 3 identical channels
    sil = (iuv255(seed, "src_sil", B, S)[..., 0] > 0).astype(np.float32)
    d["src_mask_in_image0"] = np.repeat(sil[:, None], 3, 1)
    d["tgt_img"] = uniform(seed, "tgt_img", (B, 3, S, S))
    d["tgt_IUV255"] = iuv255(seed, "tgt_iuv", B, S)
    d["tgt_IUV"] = ((d["tgt_IUV255"].astype(np.float32) / 255.0 - 0.5) * 2.0).transpose(0, 3, 1, 2).copy()
    d["bg_noise"] = normal(seed, "bg_noise", (B, 3, S, S))
    d["smpl_real_mask"] = np.repeat((d["tgt_IUV255"][..., 0] > 0).astype(np.float32)[:, None], 3, 1)
    d["tgt_verts"] = posed_vertices(seed, "tgt_verts", B)
    d["src_verts"] = posed_vertices(seed, "src_verts", B)
    cam = np.zeros((B, 3), np.float32); cam[:, 0] = 0.9
    d["tgt_cam"] = cam.copy()
    d["src_cam"] = cam.copy()
    # SMPL pose of every reference frame (smpl_vertices[:, 1 + t], train/4...py:263-266): the propagation source
    # `prosrc` picks one of them; reference 0 is `src_verts` itself
    d["src_verts_refs"] = np.stack([d["src_verts"]] + [posed_vertices(seed, "src_verts_r%d" % t, B) for t in range(1, T)], 1)
    d["src_cam_refs"] = np.tile(cam[:, None], (1, T, 1))
    d["face_bbox"] = np.tile(np.array([[96, 160, 32, 96]], np.int64), (B, 1))   # x0, x1, y0, y1
    return d


def stage4_clip(seed: int, B: int, F: int, T: int = 4, S: int = 256) -> Dict[str, np.ndarray]:
    """B synthetic clips of F target frames for the forward-only loop (test/conv_pro_test.py:219-279,
    BASELINE config 2): the reference tensors of `stage4_batch` plus per-frame targets
    (tgt_IUV255 [B,F,S,S,3], tgt_IUV / smpl_real_mask [B,F,3,S,S], tgt_verts [B,F,NV,3], tgt_cam [B,F,3])
    and `chosen_frame` [T]: the clip positions of the T reference frames (host integers, :256-262)."""
    d = stage4_batch(seed, B, T, S)
    for k in ("tgt_img", "tgt_IUV255", "tgt_IUV", "smpl_real_mask", "tgt_verts", "tgt_cam", "face_bbox", "src_verts_refs",
              "src_cam_refs"):
        d.pop(k)
    iuv = np.stack([iuv255(seed, "clip_iuv_f%d" % f, B, S) for f in range(F)], 1)
    d["tgt_IUV255"] = iuv
    d["tgt_IUV"] = ((iuv.astype(np.float32) / 255.0 - 0.5) * 2.0).transpose(0, 1, 4, 2, 3).copy()
    d["smpl_real_mask"] = np.repeat((iuv[..., 0] > 0).astype(np.float32)[:, :, None], 3, 2)
    d["tgt_verts"] = np.stack([posed_vertices(seed, "clip_verts_f%d" % f, B) for f in range(F)], 1)
    cam = np.zeros((B, F, 3), np.float32); cam[..., 0] = 0.9
    d["tgt_cam"] = cam
    d["chosen_frame"] = np.array([(t * max(F - 1, 1)) // max(T - 1, 1) for t in range(T)], np.int64)
    return d



def stage1_batch(seed: int, B: int, T: int = 4, Z: int = 3) -> Dict[str, np.ndarray]:
    """BASELINE config 1 inputs: reference atlases and masks, Z target atlases and masks
    (Fusion_dataset_textonly after the permutes of train/1.text_accu_LSTM.py:121-127)."""
    return {"src_texture_im": uniform(seed, "s1_src_tex", (B, T, 3, 800, 1200)),
            "src_mask_im": rect_masks(seed, "s1_src_mask", (B, T, 800, 1200)),
            "tgt_texture_im": uniform(seed, "s1_tgt_tex", (B, Z, 3, 800, 1200)),
            "tgt_mask_im": rect_masks(seed, "s1_tgt_mask", (B, Z, 800, 1200))}


def stage4_raw(seed: int, B: int, T: int = 4, S: int = 256) -> Dict[str, np.ndarray]:
    """A stage-4 sample batch as the dataset holds it right after decoding (uint8, HWC, BGR order is irrelevant here):
    input of jafpro_amd.data.stage4_batch_from_uint8 / oracle.data_oracle.stage4_batch."""
    r = _rng(seed, "raw")
    d: Dict[str, np.ndarray] = {}
    d["src_texture_u8"] = r.integers(0, 256, (B, T, 800, 1200, 3), dtype=np.uint8)
    d["src_mask_u8"] = (rect_masks(seed, "raw_mask", (B, T, 800, 1200)) * 255).astype(np.uint8)
    d["src_img_u8"] = r.integers(0, 256, (B, T, S, S, 3), dtype=np.uint8)
    d["tgt_img_u8"] = r.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    d["src_IUV0_u8"] = iuv255(seed, "raw_src_iuv", B, S)
    d["tgt_IUV_u8"] = iuv255(seed, "raw_tgt_iuv", B, S)
    d["smpl_real_mask_u8"] = np.repeat(((d["tgt_IUV_u8"][..., 0] > 0) * 255).astype(np.uint8)[..., None], 3, -1)
    d["smpl_real_mask_u8"][:, ::7, ::5] = 131                 # soft edges: values other than 0 / 255
    return d


def uv_assets(dirpath: str, seed: int = 0, faces: np.ndarray = None) -> Dict[str, str]:
    """Synthetic stand-ins for the reference's UV-map assets (`mapper.txt` + three JSON face lists, SURVEY F12: not
    redistributable), in the same file formats, for a given [F, 3] topology (default: a 12 x 9 grid of quads = 216 triangles):
    an OBJ with `v`, `vn`, `vt` and `f v/vt/vn` records whose texture coordinates are seeded and partly OUTSIDE the unit square
    (the sampler clips), a partition of the faces into 10 named parts, a `head` list and a `front` list inside it.
    Returns {"obj", "part_info", "front_info", "head_info", "nf"}."""
    import json
    import os
    r = _rng(seed, "uv_assets")
    if faces is None:
        gx, gy = 12, 9
        idx = lambda i, j: i * (gy + 1) + j
        quads = [(idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)) for i in range(gx) for j in range(gy)]
        faces = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], dtype=np.int64)
    faces = np.asarray(faces, dtype=np.int64)
    nf, nv = int(faces.shape[0]), int(faces.max()) + 1
    verts = r.uniform(-1, 1, (nv, 3)).astype(np.float32)
    # one texture vertex per face corner (UV seams everywhere), a few of them shared again
    nvt = 3 * nf
    vts = r.uniform(-0.05, 1.05, (nvt, 2)).astype(np.float32)
    fvt = np.arange(nvt, dtype=np.int64).reshape(nf, 3)
    share = r.integers(0, nf, size=max(1, nf // 8))
    fvt[share, 0] = fvt[(share + 1) % nf, 1]
    os.makedirs(dirpath, exist_ok=True)
    obj = os.path.join(dirpath, "mapper.txt")
    with open(obj, "w") as fp:
        for v in verts:
            fp.write("v %.6f %.6f %.6f\n" % (v[0], v[1], v[2]))
        for v in verts:
            fp.write("vn %.6f %.6f %.6f\n" % (v[2], v[0], v[1]))
        for t in vts:
            fp.write("vt %.6f %.6f\n" % (t[0], t[1]))
        for f, t in zip(faces, fvt):
            fp.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (f[0] + 1, t[0] + 1, f[0] + 1, f[1] + 1, t[1] + 1, f[1] + 1,
                                                        f[2] + 1, t[2] + 1, f[2] + 1))
    part_of = r.integers(0, 10, size=nf)
    part_of[:10] = np.arange(10)                     # no empty part
    parts = {"part_%02d" % p: {"face": [int(i) for i in np.nonzero(part_of == p)[0]]} for p in range(10)}
    head = sorted(int(i) for i in r.choice(nf, size=max(4, nf // 5), replace=False))
    front = sorted(int(i) for i in r.choice(head, size=max(2, len(head) // 2), replace=False))
    paths = {"obj": obj, "part_info": os.path.join(dirpath, "smpl_part_info.json"),
             "front_info": os.path.join(dirpath, "front_facial.json"), "head_info": os.path.join(dirpath, "head.json")}
    json.dump(parts, open(paths["part_info"], "w"))
    json.dump({"face": front}, open(paths["front_info"], "w"))
    json.dump({"face": head}, open(paths["head_info"], "w"))
    paths["nf"] = nf
    return paths
