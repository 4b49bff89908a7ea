"""UV-map assets of SMPLRenderer (the static-UV branch): the host-side builders the reference runs once in
`SMPLRenderer.__init__` (src/nmr.py:144-159) on its UV OBJ `mapper.txt` and three JSON face lists.  The assets themselves are
not redistributable and are not shipped; given by the caller (same file formats), these functions build the same
buffers:

  load_obj            src/mesh.py:28-77     `v` / `vn` / `vt` / `f a/b/c` records
  create_uvsampler    src/mesh.py:530-568   [F, T*T, 2] sampling grid of every face's T x T texels, in [-1, 1]
  get_f2vts           src/mesh.py:173-194   [F, 3, 3] UV corners (v flipped, z = 0), optionally with the back faces
  compute_barycenter  src/mesh.py:156-170
  create_mapping      src/mesh.py:368-423   per-face encodings `uv`, `seg`, `uv_seg`, `par`, `front`, `head`, `back`, `ids`,
                                            `binary` (+ the background row)

NumPy float32 throughout, as the reference: the results are pinned bit for bit on the imported reference module
(oracle/make_golden.py g_mesh -> tests/golden/mesh_assets.npz, tests/test_host_logic.py).
"""
from __future__ import annotations

import itertools
import json

import numpy as np


def load_obj(obj_file):
    """{vertices, faces, vts, vns, faces_vts, faces_vns}: every `f` record must carry the three indices `v/vt/vn`; any other
    record type raises (src/mesh.py:28-77)."""
    rec = {"v": [], "vn": [], "vt": []}
    tri = [[], [], []]                      # vertex / texture / normal indices of the faces
    with open(obj_file, "r") as fp:
        for line in fp:
            parts = line.rstrip().split()
            kind = parts[0]
            if kind in ("v", "vn"):
                rec[kind].append(parts[1:4])
            elif kind == "vt":
                rec[kind].append(parts[1:3])
            elif kind == "f":
                corners = [p.split("/") for p in parts[1:4]]
                for k in range(3):
                    tri[k].append([c[k] for c in corners])
            else:
                raise ValueError(kind)
    f32 = lambda rows, w: np.array(rows, dtype=np.float32).reshape(-1, w) if rows else np.array([], dtype=np.float32)
    idx = lambda rows: (np.array(rows, dtype=np.int32) - 1) if rows else np.array([], dtype=np.int32)
    return {"vertices": f32(rec["v"], 3), "faces": idx(tri[0]), "vts": f32(rec["vt"], 2), "vns": f32(rec["vn"], 3),
            "faces_vts": idx(tri[1]), "faces_vns": idx(tri[2])}


def _uv_corners(uv_mapping_path):
    """(vts with v flipped [NVT, 2], faces_vts [F, 3])."""
    obj = load_obj(uv_mapping_path)
    vts = obj["vts"]
    vts[:, 1] = 1 - vts[:, 1]
    return vts, obj["faces_vts"]


def compute_barycenter(f2vts):
    """[F, 3, C] -> [F, C]: v2 + (v0 - v2) / 2 + (v1 - v2) / 2, in the reference's order of operations."""
    v2 = f2vts[:, 2]
    return v2 + 0.5 * (f2vts[:, 0] - f2vts[:, 2]) + 0.5 * (f2vts[:, 1] - f2vts[:, 2])


def get_f2vts(uv_mapping_path, fill_back=False):
    vts, faces = _uv_corners(uv_mapping_path)
    vts = np.concatenate([vts, np.zeros((vts.shape[0], 1), dtype=np.float32)], axis=-1)
    if fill_back:
        faces = np.concatenate((faces, faces[:, ::-1]), axis=0)
    return vts[faces]


def create_uvsampler(uv_mapping_path="data/uv_mappings.txt", tex_size=2):
    """[F, T*T, 2]: texel (i, j) of a face sits at v2 + a_i (v0 - v2) + b_j (v1 - v2) with a, b on the T-point grid of [0, 1]
    (the order neural_renderer walks a face's texture), clipped to the unit square, mapped to [-1, 1]."""
    grid = np.arange(tex_size, dtype=np.float32) / (tex_size - 1)
    coords = np.stack([p for p in itertools.product(*[grid, grid])])             # [T*T, 2]
    vts, faces = _uv_corners(uv_mapping_path)
    f2vts = vts[faces]                                                            # [F, 3, 2]
    v2 = f2vts[:, 2]
    edges = np.dstack([f2vts[:, 0] - f2vts[:, 2], f2vts[:, 1] - f2vts[:, 2]])     # [F, 2 (x, y), 2 (edge)]
    samples = edges.dot(coords.T) + v2.reshape(-1, 2, 1)                          # [F, 2, T*T]
    samples = np.clip(samples, a_min=0.0, a_max=1.0)
    return np.transpose(samples, (0, 2, 1)) * 2 - 1


def _face_list(path, key="face"):
    with open(path, "r") as reader:
        return json.load(reader)[key]


def _with_back(faces, nf, fill_back):
    return faces + [f + nf // 2 for f in faces] if fill_back else faces


def _mark(nf, faces):
    m = np.zeros((nf, 1), dtype=np.float32)
    m[faces] = 1.0
    return m, np.zeros((1, 1), dtype=np.float32)


def create_mapping(map_name, mapping_path="../mapper.txt", part_info="../smpl_part_info.json",
                   front_info="../front_facial.json", head_info="../head.json", contain_bg=True, fill_back=False):
    """[F (+1), C] per-face encoding; with `contain_bg` the background row is appended (face index -1 selects it)."""
    f2vts = get_f2vts(mapping_path, fill_back=fill_back)
    nf = f2vts.shape[0]
    if map_name == "uv":
        map_fn, bg = compute_barycenter(f2vts)[:, 0:2], np.array([[-1, -1]], dtype=np.float32)
    elif map_name == "seg":
        map_fn, bg = np.ones((nf, 1), dtype=np.float32), np.array([[0]], dtype=np.float32)
    elif map_name == "uv_seg":
        map_fn, bg = compute_barycenter(f2vts), np.array([[0, 0, 1]], dtype=np.float32)
    elif map_name == "par":
        # NB the reference does not forward fill_back here (src/mesh.py:402): kept
        with open(part_info, "r") as reader:
            part_data = json.load(reader)
        ndim = len(part_data) + 1
        map_fn = np.zeros((nf, ndim), dtype=np.float32)
        seen = set()
        for i, name in enumerate(sorted(part_data.keys())):
            faces = part_data[name]["face"]
            map_fn[faces, i] = 1.0
            seen |= set(faces)
        assert len(seen) == nf, "nf_counter = {}, nf = {}".format(len(seen), nf)
        bg = np.zeros((1, ndim), dtype=np.float32)
        bg[0, -1] = 1
    elif map_name == "front":
        map_fn, bg = _mark(nf, _with_back(_face_list(front_info), nf, fill_back))
    elif map_name == "head":
        map_fn, bg = _mark(nf, _with_back(_face_list(head_info), nf, fill_back))
    elif map_name == "back":
        faces = list(set(_face_list(head_info)) - set(_face_list(front_info)))
        map_fn, bg = _mark(nf, _with_back(faces, nf, fill_back))
    elif map_name == "ids":
        map_fn, bg = np.arange(0, 1, 1 / nf, dtype=np.float32), np.array([[-1]], dtype=np.float32)
    elif map_name == "binary":
        width = len(np.binary_repr(nf))
        map_fn = np.stack([np.array(list(map(int, np.binary_repr(i, width=width)))) for i in range(nf)], axis=0)
        bg = np.zeros((1, width), dtype=np.float32) - 1.0
    else:
        raise ValueError("map name error {}".format(map_name))
    if contain_bg:
        map_fn = np.concatenate([map_fn, bg], axis=0)
    return map_fn
