"""ORACLE PIN (survey container only: reads /root/reference read-only; never runs on the GPU box).

Checks oracle/raster_oracle.c + the oracle's look_at / perspective against the reference's own
golden vectors for the rasteriser path:
  * tests/test_rasterize_silhouettes.py:16-35  teapot silhouette == tests/data/teapot_blender.png
  * tests/test_look_at.py:9-25                 look_at known answers
  * tests/test_perspective.py:9-14             perspective known answer
and writes tests/golden/raster_pin.json with the verdicts plus a small fixture of OUR OWN
(single triangle / tetrahedron-like / procedural body mesh digests) that the CPU and GPU tests
replay.  No reference file content is copied into the repo.
"""
import json
import os
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import raster_oracle, torch_oracle as O   # noqa: E402

REF = "/root/reference/third_party/neural_renderer/tests/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "raster_pin.json")


def load_obj(path):
    """Restates neural_renderer/load_obj.py:105-147 (v / f lines, fan triangulation, unit-cube normalisation)."""
    vs, fs = [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            vs.append([float(v) for v in t[1:4]])
        elif t[0] == "f":
            idx = [int(v.split("/")[0]) for v in t[1:]]
            for i in range(len(idx) - 2):
                fs.append((idx[0], idx[i + 1], idx[i + 2]))
    v = torch.from_numpy(np.vstack(vs).astype(np.float32))
    f = torch.from_numpy(np.vstack(fs).astype(np.int32)) - 1
    v -= v.min(0)[0][None, :]
    v /= torch.abs(v).max()
    v *= 2
    v -= v.max(0)[0][None, :] / 2
    return v, f


def main():
    res = {}
    # look_at KATs
    eyes = [[1, 0, 1], [0, 0, -10], [-1, 1, 0]]
    answers = [[-np.sqrt(2) / 2, 0, np.sqrt(2) / 2], [1, 0, 10], [0, np.sqrt(2) / 2, 3. / 2. * np.sqrt(2)]]
    v = torch.tensor([[[1., 0., 0.]]])
    res["look_at_kat"] = bool(all(np.allclose(O.look_at(v, e).squeeze().numpy(), np.array(a)) for e, a in zip(eyes, answers)))
    res["perspective_kat"] = bool(np.allclose(O.perspective(torch.tensor([[[1., 2., 10.]]])).squeeze().numpy(),
                                              np.array([np.sqrt(3) / 10, 2 * np.sqrt(3) / 10, 10], np.float32)))
    # teapot silhouette (Renderer(camera_mode='look_at'), fill_back, perspective 30deg, no anti-aliasing)
    verts, faces = load_obj(os.path.join(REF, "teapot.obj"))
    faces2 = torch.cat((faces, faces[:, [2, 1, 0]]), 0)
    vv = O.perspective(O.look_at(verts[None], [0, 0, O.EYE_Z]))
    f33 = vv[0][faces2.long()][None].numpy()
    fim, wim = raster_oracle.rasterize_fim_wim(f33, 256, 0.1, 100.0)
    sil = (fim[0] >= 0).astype(np.float32)
    ref = np.asarray(Image.open(os.path.join(REF, "teapot_blender.png")))
    ref = (ref.min(-1) != 255).astype(np.float32)
    res["teapot_silhouette_equal"] = bool(np.allclose(ref, sil))
    res["teapot_mismatch_pixels"] = int((ref != sil).sum())
    res["teapot_coverage"] = float(sil.mean())
    json.dump(res, open(OUT, "w"), indent=1)
    print(res)
    assert res["look_at_kat"] and res["perspective_kat"] and res["teapot_silhouette_equal"], res


if __name__ == "__main__":
    main()
