"""The stage-4 caller loop, jafpro_amd/train.py::run_stage4 (train/4.convLSTM_flowpro_interval.py:201-203, 249-261, 515-544):
subset / propagation-source draws, the clip kept in flight ahead of the step, the checkpoint cadence -- and BASELINE configs[1]
at its real clip length (30 target frames) against a fixture (VERDICT r4 missing 3 / weak 10)."""
import os

import numpy as np
import pytest
import torch

from tests._step_util import LOSSES, build, check_step_golden, golden_step

pytestmark = pytest.mark.gpu

# the (used, prosrc) cases the oracle's fixtures exist for (tests/test_gpu_step_parity.py), T = 1, 2, 3, 4
DRAWS = [((2,), 2), ((3, 0), 3), ((1, 2, 3), 2), ((0, 1, 2, 3), 0)]
CASES = ["s330_u2_p2", "s330_u30_p3", "s330_u123_p2", "s330_u0123_p0"]
SUBSET_GRAD_BARS = {"accu": 3e-2, "inpaint": 2e-2, "refine": 1e-2, "flow": 5e-3, "D": 5e-3, "face": 5e-3}


def test_run_stage4_iterations_match_the_subset_fixtures(tmp_path):
    """Four iterations of the loop over one clip with the draw sequence T = 1, 2, 3, 4: iteration k -- trained through the loop,
    with iteration k+1's clip prepared on the side stream under its backward pass and the plan / pack / image caches of the
    previous T still warm -- must reproduce the oracle's fixture of that subset (frame, losses, gradients of all six modules,
    BatchNorm buffers).  The fixtures are first steps from the pristine weights, so the test's callback puts the training state
    back after every iteration (Stage4Trainer.snapshot / restore); count runs 12001.. and the checkpoint cadence writes the
    seven files the script writes, loadable into fresh modules."""
    from jafpro_amd import train
    from jafpro_amd.stages import checkpoint_path, load_checkpoint
    M, tr, _, batch, dbatch, mods = build(1, seed=330)
    snap = tr.snapshot()
    seen = []

    def on_step(count, out, used, prosrc, b):
        k = len(seen)
        assert (used, prosrc) == DRAWS[k] and count == train.START_COUNT + 1 + k
        check_step_golden(CASES[k], out, mods, grad_bars=SUBSET_GRAD_BARS, bn_tol=1e-4, tag="iteration %d %s" % (k, CASES[k]))
        seen.append(count)
        if k < 3:                       # (the last iteration's update stays: it is what the checkpoint below must hold)
            tr.restore(snap)

    hist = train.run_stage4(tr, [dbatch] * 4, draws=DRAWS, ckpt_dir=str(tmp_path), save_interval=4, on_step=on_step)
    torch.cuda.synchronize()
    assert [h["count"] for h in hist] == [12001, 12002, 12003, 12004] and [h["used"] for h in hist] == [d[0] for d in DRAWS]
    assert all(bool(torch.isfinite(h[k]).all()) for h in hist for k in LOSSES)
    # 12004 % 4 == 0: the seven files of train/4...py:518-533, holding the weights AFTER iteration 12004's update
    names = sorted(os.listdir(tmp_path))
    assert names == sorted("%s_iter_12004.pth" % p for p in ("Accu", "inpaint", "bg", "refine", "D", "FD", "pro")), names
    from tests._step_util import gpu_models
    M2, mods2 = gpu_models()
    for n in ("accu", "inpaint", "bg", "refine", "D", "face", "flow"):
        load_checkpoint(mods2[n], checkpoint_path(str(tmp_path), n, 12004))
        a, b = mods[n].state_dict(), mods2[n].state_dict()
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a), n
    assert not torch.equal(tr.flat["accu"].flat, snap[0]["accu"][0])            # ... which did move
    assert all(m.training for m in (M.Accu_model, M.inpaint_model, M.bg_model, M.refine_model, M.propagater))    # :536-542


def test_run_stage4_seeded_draws_and_uint8_loader():
    """seed=16 draws T = 1, 2, 3, 4 in its first four iterations (the stream of np.random.seed(16) under the script's calls,
    tests/test_host_logic.py pins draw_subset against them); the loader yields the dataset's uint8 form, which the loop sends
    through the device input pipeline (data.stage4_batch_from_uint8); every iteration is given the next clip.  bf16 arithmetic:
    the benchmarked mode under changing T (plan / image caches keyed by geometry)."""
    from jafpro_amd import ops, synth, train
    M, tr, _, _, _, _ = build(1)
    raws = []
    for i in range(5):
        raw = synth.stage4_raw(770 + i, 2)
        b = synth.stage4_batch(770 + i, 2)
        for k in ("src_verts_refs", "src_cam_refs", "tgt_verts", "tgt_cam", "src_verts", "src_cam"):
            raw[k] = b[k]
        raw["face_bbox"] = b["face_bbox"]
        raws.append(raw)
    rng = np.random.RandomState(16)
    want = [train.draw_subset(rng) for _ in range(5)]
    assert [len(u) for u, _ in want[:4]] == [1, 2, 3, 4]
    prev = ops.set_precision("bf16")
    try:
        hist = train.run_stage4(tr, raws, seed=16)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    assert [(h["used"], h["prosrc"]) for h in hist] == want
    assert all(bool(torch.isfinite(h[k]).all()) for h in hist for k in LOSSES)
    assert tr.flat["accu"].step_count == 5 and tr.flat["D"].step_count == 15


def test_forward_clip_30_frames():
    """BASELINE configs[1] at the clip length the metric names: one clip, 30 target frames, every frame propagated from the
    reference nearest in time (test/conv_pro_test.py:256-268).  Fixture clip30_s401 (oracle/make_step_golden.py::case_clip30):
    every 16th element of every frame, per-frame sums, and frames 0 / 14 / 29 whole.  f32 and bf16x3, bar 1e-3 L-inf."""
    from jafpro_amd import ops, synth
    from jafpro_amd.step import forward_clip, _to_dev
    M, tr, _, _, _, _ = build(1)
    g = golden_step("clip30_s401")
    Fn, stride = int(g["meta.F"]), int(g["meta.stride"])
    clip = synth.stage4_clip(int(g["meta.seed"]), int(g["meta.B"]), Fn)
    assert Fn == 30 and list(clip["chosen_frame"]) == list(g["chosen_frame"]) == [0, 9, 19, 29]
    dclip = _to_dev(clip, "cuda")
    for mode in ("f32", "bf16x3"):
        prev = ops.set_precision(mode)
        try:
            out = forward_clip(M, dclip)
        finally:
            ops.set_precision(prev)
        assert out.shape == (1, 30, 3, 256, 256)
        flat = out.reshape(1, Fn, -1)
        err = (flat[:, :, ::stride].cpu() - torch.from_numpy(g["samples"])).abs().amax(-1)
        worst = float(err.max())
        print("forward_clip 30 frames (%s): sampled max|diff| %.3e (frame %d)" % (mode, worst, int(err.argmax())))
        assert worst <= 1e-3
        for f in g["meta.full"]:
            e = (out[:, int(f)].cpu() - torch.from_numpy(g["frame%d" % int(f)])).abs().max().item()
            assert e <= 1e-3, (mode, int(f), e)
        s = flat.double().sum(-1).cpu().numpy()
        assert np.abs(s - g["sum"]).max() <= 1e-3 * 3 * 65536 * 0.05, (mode, np.abs(s - g["sum"]).max())
