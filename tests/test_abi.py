"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/jafpro_hip.h
declares, rejects bad arguments before touching a GPU, and plans every layer of the stage-4 networks
within the LDS budget.  No kernel is launched here."""
import ctypes
import os

import pytest

from jafpro_amd import _lib


def test_library_exports_every_declared_symbol():
    protos = _lib.parse_header()
    assert len(protos) >= 40
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(handle, n)]
    assert not missing, missing
    assert _lib.lib().jaf_version() == 100


def test_device_code_has_no_packed_fp32_instructions(tmp_path):
    """build.py NO_PACKED_F32: v_pk_mul/add/fma_f32 gave wrong lanes beside bf16 MFMA kernels of another
    stream; the shipped code objects must not contain one.  Disassembles the gfx950 code of the built library."""
    import re
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    so = str(tmp_path / "lib.so")
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", so], check=True, capture_output=True, cwd=str(tmp_path))
    cos = [f for f in os.listdir(str(tmp_path)) if "gfx950" in f]
    assert cos, "no gfx950 code object in %s" % _lib.LIB_PATH
    n_inst, packed = 0, []
    for co in cos:
        txt = subprocess.run([objdump, "-d", str(tmp_path / co)], check=True, capture_output=True, text=True).stdout
        n_inst += len(re.findall(r"\bv_mfma_", txt))
        packed += re.findall(r"\bv_pk_(?:mul|add|fma)_f32\b", txt)
    assert n_inst > 100                      # the disassembly is the real thing
    assert not packed, "%d packed-fp32 instructions in the library" % len(packed)


def test_header_cites_reference_interfaces():
    text = open(_lib.HEADER).read()
    for cite in ("rasterize_cuda.cpp:70-95", "src/convLSTM.py:41-56", "src/crn_model.py:78-87", "src/cal_flow.py:38",
                 "train/4.convLSTM_flowpro_interval.py:43-76", "rasterize_cuda_kernel.cu:24-169"):
        assert cite in text, cite


def test_bad_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    assert L.jaf_act_bwd(None, None, None, None, 10, 1, 0.2) == -1
    assert L.jaf_avgpool_fwd(None, None, None, 1, 8, 8, 4, 4, 3, 2, 1) == -1
    assert L.jaf_adam_step(None, None, None, None, None, 16, 1e-3, 0.9, 0.999, 1e-8, 1) == -1
    assert L.jaf_texture_warp_fwd(None, None, None, None, 1, 256, 200, 200, 0) == -1
    d = _lib.ConvDesc()
    pl = _lib.ConvPlan()
    assert L.jaf_conv2d_plan(ctypes.byref(d), 0, ctypes.byref(pl)) == -1           # all-zero descriptor
    assert L.jaf_conv2d_fwd(None, ctypes.byref(d), ctypes.byref(pl), None, None, None, None, None, None) == -1
    assert L.jaf_conv2d_wgrad(None, ctypes.byref(d), None, None, None, None, None, 0) == -1
    with pytest.raises(RuntimeError):
        _lib.check(-1, "x")


def _desc(N, G, cins, Cout, H, W, k, s, p, dil=1):
    from jafpro_amd.ops import _make_desc, _out_size
    specs = [(c, G * c, 0, c) for c in cins]
    OH, OW = _out_size(H, k, s, p), _out_size(W, k, s, p)
    return _make_desc(N, G, sum(cins), Cout, H, W, OH, OW, k, k, s, p, p, dil, specs, sum(cins), 0, G * Cout, 0, 0, 0.0)


LAYERS = [  # (G, cins, Cout, H, k, s, p): SURVEY Appendix B
    (24, [3], 12, 200, 5, 1, 2), (24, [12], 24, 200, 3, 2, 1), (24, [24], 24, 100, 3, 1, 1), (24, [24], 48, 50, 3, 2, 1),
    (24, [48], 96, 25, 3, 2, 1), (24, [96], 96, 13, 3, 1, 1), (24, [96, 48], 48, 25, 3, 1, 1), (24, [12, 12], 6, 200, 3, 1, 1),
    (24, [6], 3, 200, 3, 1, 1), (24, [96, 72, 48], 96, 25, 3, 1, 1),
    (1, [3], 64, 256, 3, 1, 1), (1, [256], 256, 256, 3, 1, 1), (1, [3, 64, 512], 256, 128, 3, 1, 1), (1, [3, 512], 512, 4, 3, 1, 1),
    (1, [3, 256, 512], 512, 8, 3, 1, 1), (1, [256], 3, 256, 1, 1, 0), (1, [9], 32, 262, 7, 1, 0), (1, [32], 1, 262, 7, 1, 0),
    (1, [6], 32, 256, 3, 2, 1), (1, [128], 256, 8, 3, 2, 1), (1, [512], 512, 16, 3, 1, 1),
]


@pytest.mark.parametrize("layer", LAYERS)
def test_plan_covers_stage4_layers(layer):
    G, cins, Cout, H, k, s, p = layer
    L = _lib.lib()
    for N in (1, 8, 32):
        d = _desc(N, G, cins, Cout, H, H, k, s, p)
        pl = _lib.ConvPlan()
        assert L.jaf_conv2d_plan(ctypes.byref(d), 0, ctypes.byref(pl)) == 0
        assert pl.MT in (1, 2, 3, 4) and pl.NT in (1, 2, 4) and pl.CK % 4 == 0
        assert pl.lds_bytes <= 160 * 1024 and pl.lds_bytes == (pl.CK * pl.PS + k * k * pl.CK * pl.MRp) * 4
        assert pl.PS % 32 == 16 and pl.MRp % 32 == 16 and pl.MRp >= 16 * pl.MT
        assert pl.mblocks * 16 * pl.MT >= Cout and pl.nchunks * pl.CK >= sum(cins)
        assert pl.tiles_x * pl.tiles_p * 64 * pl.NT >= d.OH * d.OW         # every output pixel is owned by a block
        assert pl.packed_floats == G * pl.mblocks * pl.nchunks * k * k * pl.CK * pl.MRp


def test_lstm_plan_requires_gate_interleaved_tiles():
    L = _lib.lib()
    for C, H in ((12, 200), (24, 100), (24, 50), (48, 25), (96, 13)):
        d = _desc(8, 24, [C, C], 4 * C, H, H, 3, 1, 1)
        pl = _lib.ConvPlan()
        assert L.jaf_conv2d_plan(ctypes.byref(d), 1, ctypes.byref(pl)) == 0
        assert (4 * C) % (16 * pl.MT) == 0
    d = _desc(1, 1, [5, 5], 20, 8, 8, 3, 1, 1)            # 4C = 20 is not a multiple of 16: rejected
    assert L.jaf_conv2d_plan(ctypes.byref(d), 1, ctypes.byref(_lib.ConvPlan())) == -1


def test_packed_plan_invariants_and_weight_subloads():
    """jaf_conv2d_plan_packed (host code, no GPU): the LDS layout it promises the kernels, for every stage-4 layer in bf16 and
    split-bf16; and the round-5 weight sub-loads (jaf_conv_plan.pf): the wide layers with enough workgroups take 4-group chunks
    (9 exact k-steps of a 3 x 3 layer) three k-steps at a time inside 40 KB of LDS (four workgroups per CU), the small-grid layers keep
    the whole chunk resident."""
    L = _lib.lib()
    for prec in (_lib.PREC_BF16, _lib.PREC_BF16X3):
        sb = 2 if prec == _lib.PREC_BF16X3 else 1
        for (G, cins, Cout, H, k, s, p) in LAYERS:
            for N in (1, 8, 32):
                d = _desc(N, G, cins, Cout, H, H, k, s, p)
                d.precision = prec
                pl = _lib.ConvPlan()
                assert L.jaf_conv2d_plan_packed(ctypes.byref(d), 0, ctypes.byref(pl)) == 0, (G, cins, Cout, H, k, s)
                groups = (sum(cins) + 7) // 8
                assert 1 <= pl.NG <= 4 and pl.nchunks == -(-groups // pl.NG) and pl.ng_last == groups - (pl.nchunks - 1) * pl.NG
                assert pl.nsteps == -(-(k * k * pl.NG) // 4) and 0 <= pl.pf <= pl.nsteps
                wl = pl.pf if pl.pf else pl.nsteps
                assert pl.lds_bytes >= sb * pl.NG * pl.plane + sb * wl * pl.MT * 1024 + 2 * 16 * pl.nsteps * 4
                assert pl.lds_bytes <= 160 * 1024 and pl.plane % 1024 == 0 and pl.plane >= pl.npos * 16
                assert pl.mblocks * 16 * pl.MT >= Cout and pl.tiles_x * pl.tiles_p * 64 * pl.NT >= d.OH * d.OW
    # bf16: the CRN's 256 -> 256 at 256 x 256 (8192 workgroups) vs its 3 + 512 + 256 -> 512 at 32 x 32 (256 workgroups)
    d = _desc(8, 1, [256], 256, 256, 256, 3, 1, 1); d.precision = 1
    pl = _lib.ConvPlan()
    assert L.jaf_conv2d_plan_packed(ctypes.byref(d), 0, ctypes.byref(pl)) == 0
    assert (pl.MT, pl.NT, pl.NG, pl.nsteps, pl.pf) == (4, 4, 4, 9, 3) and pl.lds_bytes <= 40 * 1024
    d = _desc(8, 1, [3, 512, 256], 512, 32, 32, 3, 1, 1); d.precision = 1
    assert L.jaf_conv2d_plan_packed(ctypes.byref(d), 0, ctypes.byref(pl)) == 0
    assert pl.NG == 4 and pl.pf == 0
    # the ConvLSTM plans keep gate-interleaved tiles
    for C, H in ((12, 200), (24, 100), (24, 50), (48, 25), (96, 13)):
        d = _desc(8, 24, [C, C], 4 * C, H, H, 3, 1, 1); d.precision = 1
        assert L.jaf_conv2d_plan_packed(ctypes.byref(d), 1, ctypes.byref(pl)) == 0
        assert (4 * C) % (16 * pl.MT) == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(os.path.dirname(_lib.LIB_PATH), "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.lib()
