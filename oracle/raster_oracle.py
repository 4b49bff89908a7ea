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
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", SO, src])
    return SO


_lib = None


def rasterize_fim_wim(faces: np.ndarray, image_size: int = 256, near: float = 0.1, far: float = 100.0):
    """faces float32 [B,NF,3,3] -> (fim int32 [B,S,S], wim float32 [B,S,S,3])."""
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.raster_oracle_fim_wim.restype = ctypes.c_int
        _lib.raster_oracle_fim_wim.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    faces = np.ascontiguousarray(faces, np.float32)
    B, NF = faces.shape[0], faces.shape[1]
    fim = np.empty((B, image_size, image_size), np.int32)
    wim = np.empty((B, image_size, image_size, 3), np.float32)
    rc = _lib.raster_oracle_fim_wim(faces.ctypes.data, fim.ctypes.data, wim.ctypes.data, B, NF, image_size, near, far)
    if rc != 0:
        raise RuntimeError("raster oracle failed")
    return fim, wim
