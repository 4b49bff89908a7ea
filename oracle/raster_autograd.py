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


# ------------------------------------------------------------------------------------------------
# texture branch: RasterizeFunction with return_rgb, lighting, Renderer.render, SMPLRenderer.render
# ------------------------------------------------------------------------------------------------
class RasterizeRGBFunction(torch.autograd.Function):
    """RasterizeFunction.forward / backward with return_rgb (rasterize.py:23-160): face-index map, texture sampling,
    background fill; backward = backward_pixel_map on (rgb [, alpha]) + backward_textures [+ backward_depth_map]."""

    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, background_color, return_alpha, return_depth):
        f = faces.detach().numpy().astype(np.float32)
        t = textures.detach().contiguous().numpy().astype(np.float32)
        fim, wim, depth, finv = raster_oracle.rasterize_maps(f, image_size, near, far, flip=False)
        rgb, sidx, sw = raster_oracle.texture_sampling(f, t, fim, wim, depth, background_color, eps)
        alpha = (fim >= 0).astype(np.float32)
        ctx.cfg = (eps, return_alpha, return_depth, t.shape)
        ctx.maps = (f, fim, wim, depth, finv, alpha, rgb, sidx, sw)
        return (torch.from_numpy(rgb.copy()), torch.from_numpy(alpha), torch.from_numpy(depth.copy()), torch.from_numpy(fim),
                torch.from_numpy(wim))

    @staticmethod
    def backward(ctx, g_rgb, g_alpha, g_depth, g_fim, g_wim):
        eps, return_alpha, return_depth, tshape = ctx.cfg
        f, fim, wim, depth, finv, alpha, rgb, sidx, sw = ctx.maps
        grgb = np.zeros_like(rgb) if g_rgb is None else g_rgb.contiguous().numpy().astype(np.float32)
        ga = None
        if return_alpha:
            ga = np.zeros_like(alpha) if g_alpha is None else g_alpha.contiguous().numpy().astype(np.float32)
        g = raster_oracle.backward_pixel_map(f, fim, alpha_map=alpha if return_alpha else None, grad_alpha_map=ga,
                                             rgb_map=rgb, grad_rgb_map=grgb, eps=eps)
        gt = raster_oracle.backward_textures(fim, sw, sidx, grgb, f.shape[1], tshape[2])
        if return_depth:
            gd = np.zeros_like(depth) if g_depth is None else g_depth.contiguous().numpy().astype(np.float32)
            g = raster_oracle.backward_depth_map(f, depth, fim, finv, wim, gd, np.ascontiguousarray(g))
        return torch.from_numpy(g), torch.from_numpy(gt), None, None, None, None, None, None, None


def rasterize(faces, textures, image_size=256, anti_aliasing=True, near=0.1, far=100.0, eps=1e-4, background_color=(0, 0, 0)):
    """neural_renderer.rasterize (rasterize.py:361-391 -> rasterize_rgbad :257-358): RGB [B,3,S,S], flipped, 2x
    super-sampled + average-pooled when anti_aliasing."""
    S = image_size * 2 if anti_aliasing else image_size
    rgb, _, _, _, _ = RasterizeRGBFunction.apply(faces, textures, S, near, far, eps, background_color, False, False)
    rgb = torch.flip(rgb.permute(0, 3, 1, 2), dims=(2,))
    if anti_aliasing:
        rgb = torch.nn.functional.avg_pool2d(rgb, kernel_size=(2, 2))
    return rgb


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """neural_renderer/lighting.py:6-58 (out of place here; the reference multiplies `textures` in place)."""
    bs, nf = faces.shape[:2]
    ca = torch.as_tensor(color_ambient, dtype=torch.float32).reshape(-1, 3)
    cd = torch.as_tensor(color_directional, dtype=torch.float32).reshape(-1, 3)
    dr = torch.as_tensor(direction, dtype=torch.float32).reshape(-1, 3)
    light = torch.zeros(bs, nf, 3, dtype=torch.float32)
    if intensity_ambient != 0:
        light = light + intensity_ambient * ca[:, None, :]
    if intensity_directional != 0:
        fl = faces.reshape((bs * nf, 3, 3))
        v10 = fl[:, 0] - fl[:, 1]
        v12 = fl[:, 2] - fl[:, 1]
        normals = torch.nn.functional.normalize(torch.cross(v10, v12, dim=1), eps=1e-5).reshape((bs, nf, 3))
        cos = torch.relu(torch.sum(normals * dr[:, None, :], dim=2))
        light = light + intensity_directional * (cd[:, None, :] * cos[:, :, None])
    return textures * light[:, :, None, None, None, :]


def renderer_render(vertices, faces_idx, textures, image_size=256, anti_aliasing=True, perspective=True, fill_back=True,
                    light=(0.5, 0.5, (1, 1, 1), (1, 1, 1), (0, 1, 0)), near=0.1, far=100.0, rasterizer_eps=1e-3,
                    background_color=(0, 0, 0)):
    """Renderer(camera_mode='look_at').render (renderer.py:125-160).  vertices [B,NV,3], faces_idx [NF,3],
    textures [B,NF,ts,ts,ts,3] -> RGB [B,3,S,S]."""
    fi = torch.as_tensor(np.asarray(faces_idx)).long()
    if fill_back:
        fi = torch.cat((fi, fi[:, [2, 1, 0]]), 0)
        textures = torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)
    textures = lighting(vertices[:, fi], textures, *light)
    v = O.look_at(vertices, [0, 0, O.EYE_Z])
    if perspective:
        v = O.perspective(v)
    return rasterize(v[:, fi], textures, image_size, anti_aliasing, near, far, rasterizer_eps, background_color)


def smpl_render(cam, vertices, textures, faces_idx, image_size=256, anti_aliasing=True, near=0.1, far=25.0,
                light=(1, 0, (1, 1, 1), (1, 1, 1), (0, 1, 0)), background_color=(0, 0, 0), rasterizer_eps=1e-3):
    """SMPLRenderer.render (src/nmr.py:212-244): lighting on the un-projected faces, orthographic projection with the
    y flip, look_at, rasterize."""
    fi = torch.as_tensor(np.asarray(faces_idx)).long()
    textures = lighting(vertices[:, fi], textures.clone(), *light)
    faces = O.project_faces(vertices, cam, np.asarray(faces_idx))
    return rasterize(faces, textures, image_size, anti_aliasing, near, far, rasterizer_eps, background_color)


def create_coords(tex_size=3):
    """SMPLRenderer.create_coords (src/nmr.py:479-495): [2, T*T] barycentric sample positions."""
    step = 1 if tex_size == 1 else 1 / (tex_size - 1)
    ab = torch.arange(0, 1 + step, step, dtype=torch.float32)
    xv, yv = torch.meshgrid([ab, ab], indexing="ij")
    return torch.stack([xv.flatten(), yv.flatten()], dim=0)


def dynamic_sampler(cam, vertices, faces_idx, tex_size=3):
    """SMPLRenderer.dynamic_sampler (src/nmr.py:388-395): batch_orth_proj_idrot (:445-458) -> points_to_faces ->
    points_to_sampler (:460-477): [B,NF,T*T,2] grid positions of every face's texels in the image."""
    fi = torch.as_tensor(np.asarray(faces_idx)).long()
    pts = cam[:, None, 0:1] * (vertices[:, :, :2] + cam[:, None, 1:])
    f = pts[:, fi]                                                       # [B,NF,3,2]
    v2, v0v2, v1v2 = f[:, :, 2], f[:, :, 0] - f[:, :, 2], f[:, :, 1] - f[:, :, 2]
    samples = torch.matmul(torch.stack((v0v2, v1v2), dim=-1), create_coords(tex_size)) + v2.view(-1, fi.shape[0], 2, 1)
    return torch.clamp(samples.permute(0, 1, 3, 2), min=-1.0, max=1.0)


def extract_tex(uv_img, uv_sampler, tex_size=3, align_corners=False):
    """SMPLRenderer.extract_tex (src/nmr.py:366-386): grid_sample(uv_img, sampler) -> [B,NF,T,T,T,3] (the T x T samples
    repeated along a third texture axis)."""
    nf = uv_sampler.shape[1]
    tex = torch.nn.functional.grid_sample(uv_img, uv_sampler, mode="bilinear", padding_mode="zeros", align_corners=align_corners)
    tex = tex.view(-1, 3, nf, tex_size, tex_size).permute(0, 2, 3, 4, 1)
    return tex.unsqueeze(4).repeat(1, 1, 1, 1, tex_size, 1)
