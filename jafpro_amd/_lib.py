"""ctypes binding of libjafpro_hip.so, generated from include/jafpro_hip.h.

The product path has no CPU fallback: if the library is missing this module raises at import
of the first op (``lib()``), and every op raises when handed a non-GPU tensor.
"""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# JAFPRO_HIP_LIB: another build of the same library (A/B runs of compiler flags); still no fallback of any kind
LIB_PATH = os.environ.get("JAFPRO_HIP_LIB") or os.path.join(HERE, "libjafpro_hip.so")
HEADER = os.path.join(HERE, "..", "include", "jafpro_hip.h")

_PROTO = re.compile(r"^(int|int64_t)\s+(jaf_\w+)\s*\(([^;]*?)\)\s*;", re.S | re.M)


def parse_header(path: str = HEADER):
    """Returns {name: (restype, [argtype, ...])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in _PROTO.finditer(text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a or a.startswith("jaf_stream_t"):
                    argtypes.append(ctypes.c_void_p)
                elif a.startswith("int64_t"):
                    argtypes.append(ctypes.c_int64)
                elif a.startswith("float"):
                    argtypes.append(ctypes.c_float)
                elif a.startswith("double"):
                    argtypes.append(ctypes.c_double)
                elif a.startswith("int32_t") or a.startswith("int"):
                    argtypes.append(ctypes.c_int)
                else:
                    raise ValueError("unhandled C type in %s: %r" % (name, a))
        out[name] = (ctypes.c_int64 if ret == "int64_t" else ctypes.c_int, argtypes)
    return out


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("N", ctypes.c_int32), ("G", ctypes.c_int32),
        ("Cin", ctypes.c_int32), ("Cout", ctypes.c_int32),
        ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("OH", ctypes.c_int32), ("OW", ctypes.c_int32),
        ("KH", ctypes.c_int32), ("KW", ctypes.c_int32), ("stride", ctypes.c_int32),
        ("pad_t", ctypes.c_int32), ("pad_l", ctypes.c_int32),
        ("dil_in", ctypes.c_int32),
        ("nsrc", ctypes.c_int32),
        ("src_c", ctypes.c_int32 * 3), ("src_ctot", ctypes.c_int32 * 3),
        ("src_coff", ctypes.c_int32 * 3), ("src_gstride", ctypes.c_int32 * 3),
        ("w_cin_tot", ctypes.c_int32), ("w_cin_off", ctypes.c_int32),
        ("out_ctot", ctypes.c_int32), ("out_coff", ctypes.c_int32),
        ("act", ctypes.c_int32), ("slope", ctypes.c_float),
        ("precision", ctypes.c_int32),
    ]


class ConvPlan(ctypes.Structure):
    _fields_ = [
        ("MT", ctypes.c_int32), ("NT", ctypes.c_int32), ("CK", ctypes.c_int32), ("TWIN", ctypes.c_int32),
        ("tiles_x", ctypes.c_int32), ("tiles_p", ctypes.c_int32),
        ("PH", ctypes.c_int32), ("PW", ctypes.c_int32), ("PWp", ctypes.c_int32), ("PS", ctypes.c_int32),
        ("MRp", ctypes.c_int32), ("nchunks", ctypes.c_int32), ("mblocks", ctypes.c_int32),
        ("lds_bytes", ctypes.c_int32), ("packed_floats", ctypes.c_int64),
        ("precision", ctypes.c_int32), ("NG", ctypes.c_int32), ("ng_last", ctypes.c_int32),
        ("nsteps", ctypes.c_int32), ("nsteps_last", ctypes.c_int32), ("npos", ctypes.c_int32),
        ("plane", ctypes.c_int32), ("PWp_slots_unused", ctypes.c_int32), ("ilv", ctypes.c_int32), ("pf", ctypes.c_int32),
    ]


class PackedIO(ctypes.Structure):
    """jaf_packed_io (include/jafpro_hip.h)."""
    _fields_ = [("in_ng8_tot", ctypes.c_int32), ("dst", ctypes.c_void_p), ("dst_ng8_tot", ctypes.c_int32),
                ("dst_coff", ctypes.c_int32), ("dst_img_off", ctypes.c_int32), ("dst_pad_tail", ctypes.c_int32),
                ("skip_f32", ctypes.c_int32), ("accumulate_f32", ctypes.c_int32),
                ("out2", ctypes.c_void_p), ("split_rows", ctypes.c_int32),
                ("dz_mask", ctypes.c_void_p), ("dz_mask_ng8", ctypes.c_int32), ("dz_mask_coff", ctypes.c_int32),
                ("dz_slope", ctypes.c_float), ("dz_dbias", ctypes.c_void_p),
                ("out_bf16", ctypes.c_int32), ("out2_bf16", ctypes.c_int32), ("state_bf16", ctypes.c_int32),
                ("dz_mask_split", ctypes.c_int32)]


ACT_NONE, ACT_LRELU, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
PACK_FWD, PACK_DGRAD, PACK_LSTM, PACK_DGRAD_LSTM = 0, 1, 2, 3
PREC_F32, PREC_BF16, PREC_BF16X3 = 0, 1, 2

_LIB = None


def lib():
    """Loads the HIP library (once).  Raises if it has not been built: there is no fallback."""
    global _LIB
    if _LIB is None:
        # PyTorch first: it ships its own HIP runtime, and the process must hold exactly one.  Loaded the other way round (this
        # library before the first `import torch`), the runtime /opt/rocm resolves for us is already in place when torch brings its
        # copy, and the first launch fails with hipErrorNoDevice (seen with `python __graft_entry__.py --smoke`, which builds first).
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libjafpro_hip.so is missing (%s). Build it with `python -m jafpro_amd.build`; "
                "jafpro_amd has no CPU or PyTorch fallback path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (ret, argtypes) in parse_header().items():
            fn = getattr(handle, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = argtypes
        _LIB = handle
    return _LIB


def check(rc: int, what: str) -> None:
    """C-ABI status -> RuntimeError (the reference raises RuntimeError from AT_CHECK,
    rasterize_cuda.cpp:66-68)."""
    if rc != 0:
        kind = "invalid argument" if rc == -1 else ("unsupported" if rc == -2 else "hipError %d" % rc)
        raise RuntimeError("%s failed: %s" % (what, kind))
