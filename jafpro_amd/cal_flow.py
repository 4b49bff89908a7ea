"""float_estimate on the HIP kernels.  Mirrors src/cal_flow.py:13-39: render source and target
face-index / weight maps, barycentric flow T (cal_bc_transform), grid_sample(border) warp.

Unlike the reference constructor (:14-19) nothing is loaded from ../smpl_model.pkl or
../hmr_tf2pt.pth: the HMR network it builds there is never called in forward.
The SMPL face topology (13776 x 3 int) is passed in or loaded from `face_path`.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .nmr import SMPLRenderer


class float_estimate(nn.Module):
    def __init__(self, smpl_pkl=None, hmr_model_path=None, faces=None, face_path='../smpl_faces.npy',
                 image_size=256, align_corners=False):
        super().__init__()
        self.render = SMPLRenderer(image_size=image_size, tex_size=3, has_front=True, fill_back=False,
                                   faces=faces, face_path=face_path)
        # torch>=1.3 default for F.grid_sample at src/cal_flow.py:38
        self.align_corners = align_corners

    def forward(self, src_img, src_smpl, tgt_smpl):
        src_cam, _, src_vertices, _ = src_smpl
        tgt_cam, _, tgt_vertices, _ = tgt_smpl
        if not (torch.is_grad_enabled() and (src_img.requires_grad or src_vertices.requires_grad or src_cam.requires_grad)):
            # stage 4: nothing differentiates the warp -> projection, rasterisation and one fused
            # flow + border-sample kernel; the flow field T is never written
            src_faces = self.render.project(src_cam, src_vertices)
            _, fim, wim = self.render.render_fim_wim(tgt_cam, tgt_vertices)
            return ops.flow_warp(src_img.contiguous(), src_faces, fim, wim, None, self.align_corners)
        flow = self.cal_flow(src_cam, None, src_vertices, None, tgt_cam, None, tgt_vertices, None)
        return self.warp_image(src_img, flow)

    def cal_flow(self, src_cam, src_pose, src_vertices, src_shape, tgt_cam, tgt_pose, tgt_vertices, tgt_shape):
        src_faces = self.render.project(src_cam, src_vertices)      # only the projected faces of the source mesh are used (:30)
        _, tsf_fim, tsf_wim = self.render.render_fim_wim(tgt_cam, tgt_vertices)
        # src_f2verts[..., 0:2] with y *= -1 (:30-31) is folded into the kernel
        return ops.bc_transform(src_faces, tsf_fim, tsf_wim)

    def warp_image(self, src_image, flow):
        return ops.grid_sample(src_image.contiguous(), flow, padding_border=True, align_corners=self.align_corners)
