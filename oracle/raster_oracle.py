"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/raster_oracle.c (built by
oracle/Makefile into oracle/_build/libraster_oracle.so)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libraster_oracle.so")


def build():
    src = os.path.join(HERE, "raster_oracle.c")
    if os.path.exists(SO) and os.path.getmtime(SO) >= os.path.getmtime(src):
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", SO, src, "-lm"])
    return SO


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        _lib.raster_oracle_fim_wim.restype = ci
        _lib.raster_oracle_fim_wim.argtypes = [vp, vp, vp, ci, ci, ci, cf, cf]
        _lib.raster_oracle_maps.restype = ci
        _lib.raster_oracle_maps.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, cf, cf, ci]
        _lib.raster_oracle_bwd_pixel_map.restype = ci
        _lib.raster_oracle_bwd_pixel_map.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, cf]
        _lib.raster_oracle_bwd_depth_map.restype = ci
        _lib.raster_oracle_bwd_depth_map.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, ci]
        _lib.raster_oracle_texture_fwd.restype = ci
        _lib.raster_oracle_texture_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, cf]
        _lib.raster_oracle_texture_bwd.restype = ci
        _lib.raster_oracle_texture_bwd.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci]
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data


def rasterize_maps(faces: np.ndarray, image_size: int, near: float = 0.1, far: float = 100.0, flip: bool = False):
    """faces float32 [B,NF,3,3] -> (fim int32 [B,S,S], wim [B,S,S,3], depth [B,S,S], face_inv_map [B,S,S,9]);
    flip=False gives the maps RasterizeFunction saves for backward, flip=True what rasterize_rgbad returns."""
    L = _load()
    faces = np.ascontiguousarray(faces, np.float32)
    B, NF, S = faces.shape[0], faces.shape[1], image_size
    fim = np.empty((B, S, S), np.int32)
    wim = np.empty((B, S, S, 3), np.float32)
    depth = np.empty((B, S, S), np.float32)
    finv = np.empty((B, S, S, 9), np.float32)
    if L.raster_oracle_maps(faces.ctypes.data, fim.ctypes.data, wim.ctypes.data, depth.ctypes.data, finv.ctypes.data, B, NF, S,
                            near, far, 1 if flip else 0) != 0:
        raise RuntimeError("raster oracle failed")
    return fim, wim, depth, finv


def backward_pixel_map(faces, fim, alpha_map=None, grad_alpha_map=None, rgb_map=None, grad_rgb_map=None, eps: float = 1e-4):
    """-> grad_faces [B,NF,3,3] (zeros for back faces); maps unflipped."""
    L = _load()
    faces = np.ascontiguousarray(faces, np.float32)
    fim = np.ascontiguousarray(fim, np.int32)
    B, NF, S = faces.shape[0], faces.shape[1], fim.shape[1]
    arrs = [None if a is None else np.ascontiguousarray(a, np.float32) for a in (rgb_map, alpha_map, grad_rgb_map, grad_alpha_map)]
    g = np.zeros((B, NF, 3, 3), np.float32)
    if L.raster_oracle_bwd_pixel_map(faces.ctypes.data, fim.ctypes.data, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), _ptr(arrs[3]),
                                     g.ctypes.data, B, NF, S, eps) != 0:
        raise RuntimeError("raster oracle failed")
    return g


def backward_depth_map(faces, depth, fim, face_inv_map, wim, grad_depth, grad_faces):
    """ADDS the depth gradient to grad_faces [B,NF,3,3] in place and returns it."""
    L = _load()
    faces = np.ascontiguousarray(faces, np.float32)
    B, NF, S = faces.shape[0], faces.shape[1], fim.shape[1]
    a = [np.ascontiguousarray(x, np.float32) for x in (depth, face_inv_map, wim, grad_depth)]
    fim = np.ascontiguousarray(fim, np.int32)
    assert grad_faces.dtype == np.float32 and grad_faces.flags["C_CONTIGUOUS"]
    if L.raster_oracle_bwd_depth_map(faces.ctypes.data, a[0].ctypes.data, fim.ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                                     a[3].ctypes.data, grad_faces.ctypes.data, B, NF, S) != 0:
        raise RuntimeError("raster oracle failed")
    return grad_faces


def rasterize_fim_wim(faces: np.ndarray, image_size: int = 256, near: float = 0.1, far: float = 100.0):
    """faces float32 [B,NF,3,3] -> (fim int32 [B,S,S], wim float32 [B,S,S,3])."""
    _lib = _load()
    faces = np.ascontiguousarray(faces, np.float32)
    B, NF = faces.shape[0], faces.shape[1]
    fim = np.empty((B, image_size, image_size), np.int32)
    wim = np.empty((B, image_size, image_size, 3), np.float32)
    rc = _lib.raster_oracle_fim_wim(faces.ctypes.data, fim.ctypes.data, wim.ctypes.data, B, NF, image_size, near, far)
    if rc != 0:
        raise RuntimeError("raster oracle failed")
    return fim, wim


def texture_sampling(faces, textures, fim, wim, depth, background=(0.0, 0.0, 0.0), eps: float = 1e-4):
    """forward_texture_sampling + forward_background on UNFLIPPED maps -> (rgb [B,S,S,3], sampling_index_map int32
    [B,S,S,8], sampling_weight_map [B,S,S,8]).  textures [B,NF,ts,ts,ts,3]; background [3] or [B,3]."""
    L = _load()
    faces = np.ascontiguousarray(faces, np.float32)
    textures = np.ascontiguousarray(textures, np.float32)
    B, NF, S, ts = faces.shape[0], faces.shape[1], fim.shape[1], textures.shape[2]
    bg = np.ascontiguousarray(background, np.float32)
    rgb = np.empty((B, S, S, 3), np.float32)
    sidx = np.empty((B, S, S, 8), np.int32)
    sw = np.empty((B, S, S, 8), np.float32)
    a = [np.ascontiguousarray(fim, np.int32), np.ascontiguousarray(wim, np.float32), np.ascontiguousarray(depth, np.float32)]
    if L.raster_oracle_texture_fwd(faces.ctypes.data, textures.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
                                   rgb.ctypes.data, sidx.ctypes.data, sw.ctypes.data, bg.ctypes.data, 1 if bg.ndim == 2 else 0,
                                   B, NF, S, ts, eps) != 0:
        raise RuntimeError("raster oracle failed")
    return rgb, sidx, sw


def backward_textures(fim, sampling_weight_map, sampling_index_map, grad_rgb_map, NF: int, ts: int):
    """-> grad_textures [B,NF,ts,ts,ts,3]."""
    L = _load()
    fim = np.ascontiguousarray(fim, np.int32)
    B, S = fim.shape[0], fim.shape[1]
    sw = np.ascontiguousarray(sampling_weight_map, np.float32)
    si = np.ascontiguousarray(sampling_index_map, np.int32)
    g = np.ascontiguousarray(grad_rgb_map, np.float32)
    out = np.zeros((B, NF, ts, ts, ts, 3), np.float32)
    if L.raster_oracle_texture_bwd(fim.ctypes.data, sw.ctypes.data, si.ctypes.data, g.ctypes.data, out.ctypes.data, B, NF, S, ts) != 0:
        raise RuntimeError("raster oracle failed")
    return out
