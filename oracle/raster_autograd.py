"""ORACLE -- TEST INFRASTRUCTURE ONLY.  torch.autograd wrapper of the C rasteriser restatement, composed exactly as
neural_renderer composes its CUDA extension: RasterizeFunction (rasterize.py:16-160) without the texture branch,
rasterize_silhouettes / rasterize_depth (:428-481) with the vertical flip of rasterize_rgbad (:334-338), and the
Renderer's look_at pipeline (renderer.py:75-123) used by the reference's gradient KATs."""
from __future__ import annotations

import numpy as np
import torch

from . import raster_oracle
from . import torch_oracle as O


class RasterizeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces, image_size, near, far, eps, return_alpha, return_depth):
        f = faces.detach().numpy().astype(np.float32)
        fim, wim, depth, finv = raster_oracle.rasterize_maps(f, image_size, near, far, flip=False)
        alpha = (fim >= 0).astype(np.float32)                      # forward_alpha_map (rasterize.py:188-192)
        ctx.cfg = (image_size, eps, return_alpha, return_depth)
        ctx.maps = (f, fim, wim, depth, finv, alpha)
        return (torch.from_numpy(alpha), torch.from_numpy(depth.copy()), torch.from_numpy(fim), torch.from_numpy(wim))

    @staticmethod
    def backward(ctx, g_alpha, g_depth, g_fim, g_wim):
        image_size, eps, return_alpha, return_depth = ctx.cfg
        f, fim, wim, depth, finv, alpha = ctx.maps
        g = np.zeros_like(f)
        if return_alpha:
            ga = np.zeros_like(alpha) if g_alpha is None else g_alpha.contiguous().numpy().astype(np.float32)
            g = raster_oracle.backward_pixel_map(f, fim, alpha_map=alpha, grad_alpha_map=ga, eps=eps)
        if return_depth:
            gd = np.zeros_like(depth) if g_depth is None else g_depth.contiguous().numpy().astype(np.float32)
            g = raster_oracle.backward_depth_map(f, depth, fim, finv, wim, gd, np.ascontiguousarray(g))
        return torch.from_numpy(g), None, None, None, None, None, None


def rasterize_silhouettes(faces, image_size=256, near=0.1, far=100.0, eps=1e-4):
    alpha, _, _, _ = RasterizeFunction.apply(faces, image_size, near, far, eps, True, False)
    return torch.flip(alpha, dims=(1,))


def rasterize_depth(faces, image_size=256, near=0.1, far=100.0, eps=1e-4):
    _, depth, _, _ = RasterizeFunction.apply(faces, image_size, near, far, eps, False, True)
    return torch.flip(depth, dims=(1,))


def renderer_faces(vertices, faces_idx, perspective=True, fill_back=True):
    """Renderer(camera_mode='look_at') up to the rasteriser (renderer.py:75-95): fill_back, look_at from
    eye (0, 0, -(1/tan30 + 1)), optional 30-degree perspective, vertices_to_faces.  vertices [B,NV,3], faces_idx [NF,3]."""
    fi = torch.as_tensor(np.asarray(faces_idx)).long()
    if fill_back:
        fi = torch.cat((fi, fi[:, [2, 1, 0]]), 0)
    v = O.look_at(vertices, [0, 0, O.EYE_Z])
    if perspective:
        v = O.perspective(v)
    return v[:, fi]
