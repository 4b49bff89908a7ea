"""Builds libjafpro_hip.so (gfx950) in-tree with hipcc.

Run as ``python -m jafpro_amd.build`` or through ``__graft_entry__.build()``.  hipcc
cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libjafpro_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["conv.hip", "conv_pack_weights.hip", "conv_dma.hip", "conv_dma_split.hip", "wgrad.hip", "wgrad_dma.hip", "elementwise.hip", "norm.hip", "resample.hip", "gather.hip",
           "raster.hip", "raster_bwd.hip", "raster_texture.hip", "linear.hip", "ubench.hip", "input_pipeline.hip", "metrics.hip"]
# raster.hip must keep the reference's fp32 expression trees (no FMA contraction): see its header.
EXTRA = {"raster.hip": ["-ffp-contract=off"], "raster_bwd.hip": ["-ffp-contract=off"], "raster_texture.hip": ["-ffp-contract=off"]}
# No packed-fp32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) in device code.  Measured on MI355X: with them,
# flow_warp_fwd_kernel (`v_pk_mul_f32 ... op_sel` straight after 4-byte-aligned dwordx2 gathers) lost one product in 16 adjacent
# lanes in 2-8 % of its launches while bf16 MFMA kernels of another stream shared the CUs; without them 0 of 720, and the step
# is 1 ms faster (tests/test_gpu_overlap.py; tests/test_host_logic.py checks the built code objects).
# The x86 pass of hipcc ignores the feature with a warning.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"] + NO_PACKED_F32


def _digest(path: str, flags) -> str:
    h = hashlib.sha256()
    for p in (path, os.path.join(CSRC, "jaf_common.h"), os.path.join(CSRC, "conv_internal.h"), os.path.join(CSRC, "conv_dma_kernel.h"), os.path.join(CSRC, "jaf_fdiv.h"),
              os.path.join(HERE, "..", "include", "jafpro_hip.h")):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    flags = COMMON + EXTRA.get(src, [])
    stamp = obj + ".sha"
    dig = _digest(path, flags)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [HIPCC] + flags + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(_compile, SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
