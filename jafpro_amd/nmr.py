"""SMPL renderer pieces that float_estimate uses, on the HIP kernels.  Mirrors the live part of
src/nmr.py: orthographic_proj_withz_idrot (:10-28), SMPLRenderer.render_fim_wim (:263-278) and
SMPLRenderer.cal_bc_transform (:617-659).  The texture / sampler buffers the reference builds in
__init__ (:146-159) are never read on this path and are not built.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import ops


def orthographic_proj_withz_idrot(X, cam, offset_z=0.):
    """sc * (x + [tx; ty]), z preserved (+offset).  Tiny host-side helper kept for API parity;
    render_fim_wim fuses it with look_at / vertices_to_faces in one kernel."""
    scale = cam[:, 0].contiguous().view(-1, 1, 1)
    trans = cam[:, 1:3].contiguous().view(cam.size(0), 1, -1)
    proj_xy = scale * (X[:, :, :2] + trans)
    proj_z = X[:, :, 2, None] + offset_z
    return torch.cat((proj_xy, proj_z), 2)


class SMPLRenderer(nn.Module):
    def __init__(self, face_path='../smpl_faces.npy', uv_map_path=None, map_name='uv_seg', tex_size=3,
                 image_size=256, anti_aliasing=True, fill_back=False, background_color=(0, 0, 0),
                 viewing_angle=30, near=0.1, far=25.0, has_front=False, faces=None):
        super().__init__()
        if faces is None:
            faces = np.load(face_path)
        faces = np.asarray(faces)
        if fill_back:
            faces = np.concatenate((faces, faces[:, ::-1]), axis=0)
        self.image_size = image_size
        self.fill_back = fill_back
        self.tex_size = tex_size
        self.register_buffer('faces', torch.tensor(faces.astype(np.int32)).int().contiguous())
        self.nf = int(faces.shape[0])
        self.near, self.far = near, far
        self.viewing_angle = viewing_angle
        self.eye = [0, 0, -(1. / np.tan(np.radians(self.viewing_angle)) + 1)]
        self.rasterizer_eps = 1e-3          # src/nmr.py:170 (only the textured render path passes it on)
        self.anti_aliasing = anti_aliasing

    def project(self, cam, vertices, faces=None):
        """Projection + y-flip + look_at + vertices_to_faces (src/nmr.py:269-276) -> faces [B,NF,3,3]."""
        fidx = self.faces if faces is None else faces[0].int().contiguous()
        eye_z = float(np.float32(self.eye[2]))
        return ops.project_faces(vertices.float().contiguous(), cam.float().contiguous(), fidx, eye_z)

    def render_fim_wim(self, cam, vertices, faces=None):
        """-> (faces [B,NF,3,3] after projection/look_at, fim int32 [B,S,S], wim [B,S,S,3]).
        rasterize_face_index_map_and_weight_map defaults: near 0.1, far 100, no anti-aliasing
        (src/nmr.py:277, rasterize.py:8-13)."""
        f = self.project(cam, vertices, faces)
        fim, wim = ops.rasterize_fim_wim(f, self.image_size, 0.1, 100.0)
        return f, fim, wim

    def _supersampled(self, fn, cam, vertices, faces):
        """rasterize_rgbad's anti-aliasing (rasterize.py:316-348): render at 2x, flip, 2x2 average pool."""
        f = self.project(cam, vertices, faces)
        if not self.anti_aliasing:
            return fn(f, self.image_size, 0.1, 100.0, 1e-4)
        img = fn(f, self.image_size * 2, 0.1, 100.0, 1e-4)
        return ops.avg_pool(img.unsqueeze(1).contiguous(), 2, 2, 0)[:, 0]

    def render_silhouettes(self, cam, vertices, faces=None):
        """src/nmr.py:295-310: silhouette [B,S,S] of the projected mesh via neural_renderer.rasterize_silhouettes
        (near 0.1, far 100, eps 1e-4: rasterize.py:8-13), differentiable w.r.t. vertices and camera with
        neural_renderer's approximate silhouette gradient (rasterize_cuda_kernel.cu:245-491)."""
        return self._supersampled(ops.rasterize_silhouettes, cam, vertices, faces)

    def render_depth(self, cam, vertices, faces=None):
        """Depth map [B,S,S] (far = 100 where nothing is hit).  The reference method is a stub that raises
        (src/nmr.py:279-292); the renderer it was meant to call is neural_renderer.rasterize_depth (rasterize.py:455-481)."""
        return self._supersampled(ops.rasterize_depth, cam, vertices, faces)

    def cal_bc_transform(self, src_f2pts, dst_fims, dst_wims):
        """src_f2pts: (bs, nf, 3, 2) source face vertices (x, y already re-flipped); returns T (bs,S,S,2)."""
        bs, nf = src_f2pts.shape[0], src_f2pts.shape[1]
        # the kernel takes [B,NF,3,3] faces and negates y itself; rebuild that view of the data
        full = torch.zeros((bs, nf, 3, 3), device=src_f2pts.device, dtype=torch.float32)
        full[..., 0] = src_f2pts[..., 0]
        full[..., 1] = -src_f2pts[..., 1]
        return ops.bc_transform(full, dst_fims.int().contiguous(), dst_wims.contiguous())
