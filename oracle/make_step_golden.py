"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Run in the BUILD container (CPU, ~15 min on 8 cores, needs ~45 GB for B=8):

    python -m oracle.make_step_golden [case ...]          # no argument: every case

Runs the CPU restatements of the train steps (oracle/step_oracle.py = train/4.convLSTM_flowpro_interval.py:206-413 and
its N-rank form; oracle/stage_oracle.py = train/1-3*.py; each module of them pinned bit-exactly against the imported
reference modules by oracle/make_golden.py) on the portable synthetic weights / batches of jafpro_amd/synth.py and
writes what the GPU parity tests compare against:

    tests/golden/step_index.npz       sample positions, reference key order and tensor sizes of every module
    tests/golden/step_<case>.npz      whole generated frames, the six losses, BatchNorm buffers, and per module the
                                      gradient digest of oracle/step_digest.py (per-tensor sums of squares + samples);
                                      post-Adam parameter samples where a test checks the update

The GPU box regenerates the same weights and batches from the seeds (NumPy PCG64) and never runs these oracles.
Cases = the configurations of tests/test_gpu_step.py, test_gpu_step_parity.py and test_gpu_stages.py.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import step_digest as SD                    # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
LOSSES = ("total_loss", "vgg_l1", "errD", "errG", "F_errD", "F_errG")
TRAINABLE = ("accu", "inpaint", "refine", "flow", "D", "face")
FWD_KEYS = ("accu", "inpaint", "inpaint_warp", "refine_output", "fg_mask", "bg_output", "fusion_output", "tsf_image",
            "final_mask", "final_output")


def _host(batch):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) and k != "chosen_frame" else v)
            for k, v in batch.items()}


def _trainable(sd):
    return {k: p for k, p in sd.items() if p.requires_grad}


class Index:
    """tests/golden/step_index.npz: made from the first module of each size that is seen, then only read."""

    def __init__(self):
        self.path = os.path.join(GOLD, "step_index.npz")
        self.d = dict(np.load(self.path)) if os.path.exists(self.path) else {}
        self.dirty = False

    def of(self, name, sd):
        if "idx." + name not in self.d:
            tr = _trainable(sd)
            numel = np.array([p.numel() for p in tr.values()], np.int64)
            self.d["keys." + name] = np.array(list(tr.keys()))
            self.d["numel." + name] = numel
            self.d["idx." + name] = SD.sample_index(int(numel.sum()))
            self.dirty = True
        assert list(self.d["keys." + name]) == list(_trainable(sd).keys()), name
        return self.d["idx." + name]

    def save(self):
        if self.dirty:
            np.savez_compressed(self.path, **self.d)


INDEX = Index()


def _grads(out, prefix, name, sd, index_name=None):
    idx = INDEX.of(index_name or name, sd)
    dg = SD.digest({k: p.grad.detach().numpy() for k, p in _trainable(sd).items()}, idx)
    out["%sg.%s.sq" % (prefix, name)] = dg["sq"]
    out["%sg.%s.val" % (prefix, name)] = dg["val"]


def _params(out, prefix, name, sd, index_name=None):
    idx = INDEX.of(index_name or name, sd)
    out["%sp.%s.val" % (prefix, name)] = SD.flat_of(p.detach().numpy() for p in _trainable(sd).values())[idx]


def _bn(out, prefix, name, sd):
    for k, v in sd.items():
        if "running_" in k or k.endswith("num_batches_tracked"):
            out["%sbn.%s.%s" % (prefix, name, k)] = v.detach().numpy().copy()


def _losses(r):
    return np.array([float(r[k].reshape(-1)[0]) for k in LOSSES], np.float64)


def _stage4_oracle():
    from oracle.step_oracle import OracleStage4
    from tests._step_util import build_models
    _, _, sds, fidx = build_models()
    return OracleStage4(sds, fidx)


def case_stage4(seed, B, used=(0, 1, 2, 3), prosrc=0, steps=1, params=False, S=256):
    """One (or two consecutive) single-process train steps.  S=512: BASELINE config 5 geometry as oracle/step_oracle.py
    defines it (its header: the reference itself is 256-only)."""
    from jafpro_amd import synth
    orc = _stage4_oracle()
    b = _host(synth.stage4_batch(seed, B, S=S))
    out = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.used": np.array(used, np.int64),
           "meta.prosrc": np.int64(prosrc), "meta.S": np.int64(S)}
    r = orc.train_step(b, used=used, prosrc=prosrc)
    out["final_output"] = r["final_output"].numpy().astype(np.float32)
    out["losses"] = _losses(r)
    for n in TRAINABLE:
        _grads(out, "", n, orc.sd[n])
        if params:
            _params(out, "", n, orc.sd[n])
    for n in ("flow", "D", "face"):
        _bn(out, "", n, orc.sd[n])
    if steps == 2:
        r2 = orc.train_step(b, used=used, prosrc=prosrc)
        out["step2.losses"] = _losses(r2)
        out["step2.final_output"] = r2["final_output"].numpy().astype(np.float32)
    return out


def case_forward(seed=300, B=1, S=256):
    """generator_forward of the initial weights (BASELINE config 2 chain for one target frame; S=512: config 5 geometry)."""
    from jafpro_amd import synth
    orc = _stage4_oracle()
    with torch.no_grad():
        r = orc.generator_forward(_host(synth.stage4_batch(seed, B, S=S)), (0, 1, 2, 3), 0)
    out = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.S": np.int64(S)}
    for k in FWD_KEYS:
        a = r[k].numpy().astype(np.float32)
        if a.size <= 3 * 256 * 256 * B or k == "final_output":
            out["fwd." + k] = a                                   # frames: whole
        else:
            out["fwd." + k + ".strided"] = SD.strided(a, 65536)   # the 24-part tensors (11.5 MB each): strided samples
            out["fwd." + k + ".sum"] = np.float64(a.astype(np.float64).sum())
            out["fwd." + k + ".sq"] = np.float64((a.astype(np.float64) ** 2).sum())
    return out


def case_clip(seed=400, B=2, Fn=3):
    """BASELINE config 2: forward-only clip loop (test/conv_pro_test.py:219-279)."""
    from jafpro_amd import synth
    orc = _stage4_oracle()
    clip = synth.stage4_clip(seed, B, Fn)
    ref = orc.forward_clip(_host(clip))
    return {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.F": np.int64(Fn),
            "pred_target": ref.numpy().astype(np.float32)}


def case_clip30(seed=401, B=1, Fn=30, full=(0, 14, 29), stride=16):
    """BASELINE config 2 at its real clip length: 30 target frames (the 30-frame loop was only timed).  The
    whole [B,30,3,256,256] result is 23.6 MB, so the fixture keeps every `stride`-th element of EVERY frame (the test's L-inf
    runs over those 12 288 positions per frame), the float64 sum and sum of squares of every frame, and frames `full` whole."""
    from jafpro_amd import synth
    orc = _stage4_oracle()
    clip = synth.stage4_clip(seed, B, Fn)
    ref = orc.forward_clip(_host(clip)).numpy().astype(np.float32)
    flat = ref.reshape(B, Fn, -1)
    out = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.F": np.int64(Fn), "meta.stride": np.int64(stride),
           "meta.full": np.array(full, np.int64), "samples": flat[:, :, ::stride].copy(),
           "sum": flat.astype(np.float64).sum(-1), "sq": (flat.astype(np.float64) ** 2).sum(-1),
           "chosen_frame": np.asarray(clip["chosen_frame"], np.int64)}
    for f in full:
        out["frame%d" % f] = ref[:, f].copy()
    return out


def case_ranks(seed=340, world=2, used=(0, 1, 2, 3), prosrc=1, drop_face_rank=-1):
    """N ranks == the restated step on N chunks (per-chunk BatchNorm statistics, averaged gradients)."""
    from jafpro_amd import synth
    from jafpro_amd.dist import shard_batch
    orc = _stage4_oracle()
    full = synth.stage4_batch(seed, world)
    if drop_face_rank >= 0:
        full["face_bbox"][drop_face_rank] = (96, 96, 32, 96)
    refs = orc.train_step_ranks([_host(shard_batch(full, r, world)) for r in range(world)], used, prosrc)
    views = orc._rank_views(world)
    out = {"meta.seed": np.int64(seed), "meta.world": np.int64(world), "meta.used": np.array(used, np.int64),
           "meta.prosrc": np.int64(prosrc), "meta.drop_face_rank": np.int64(drop_face_rank)}
    for r in range(world):
        out["r%d.final_output" % r] = refs[r]["final_output"].numpy().astype(np.float32)
        out["r%d.losses" % r] = _losses(refs[r])
        for n in ("flow", "D", "face"):
            _bn(out, "r%d." % r, n, views[r][n])
    for n in TRAINABLE:
        _grads(out, "", n, orc.sd[n])
        _params(out, "", n, orc.sd[n])
    return out


def case_stage12(used, seed=620):
    """train/1.text_accu_LSTM.py:116-176 and train/2.text_inpaint_convLSTM.py:118-221 on one batch."""
    from jafpro_amd import synth
    from jafpro_amd.networks import Accumulate_LSTM, Accumulate_LSTM_no_loss, UNet_inpainter
    from oracle.stage_oracle import OracleStage1, OracleStage2
    hb = _host(synth.stage1_batch(seed, 1))
    out = {"meta.seed": np.int64(seed), "meta.used": np.array(used, np.int64)}
    m1 = synth.load_synth(Accumulate_LSTM(), 121)
    o1 = OracleStage1({k: v.detach().clone() for k, v in m1.state_dict().items()})
    r = o1.train_step(hb, used)
    out["s1.total_loss"] = np.float64(float(r["total_loss"]))
    tex = r["output_texture"].numpy().astype(np.float32)
    out["s1.output_texture.strided"] = SD.strided(tex, 65536)
    out["s1.output_texture.sq"] = np.float64((tex.astype(np.float64) ** 2).sum())
    _grads(out, "s1.", "accu", o1.sd, index_name="accu")
    accu, inp = synth.load_synth(Accumulate_LSTM_no_loss(), 122), synth.load_synth(UNet_inpainter(), 123)
    o2 = OracleStage2({k: v.detach().clone() for k, v in accu.state_dict().items()},
                      {k: v.detach().clone() for k, v in inp.state_dict().items()})
    r = o2.train_step(hb, used)
    out["s2.total_loss"] = np.float64(float(r["total_loss"]))
    a = r["inpaint"].numpy().astype(np.float32)
    out["s2.inpaint.strided"] = SD.strided(a, 65536)
    out["s2.inpaint.sq"] = np.float64((a.astype(np.float64) ** 2).sum())
    for n in ("accu", "inpaint"):
        _grads(out, "s2.", n, o2.sd[n])
    return out


def case_stage3(seed=630, B=2, used=(1, 3, 0)):
    """train/3.inpaint_global_convLSTM_FGAN.py:193-382."""
    from jafpro_amd import synth
    from jafpro_amd.stages import Stage3Models
    from oracle.stage_oracle import OracleStage3
    from tests._step_util import SEEDS
    M = Stage3Models()
    mods = {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
            "D": M.discriminator, "face": M.F_Discriminator, "vgg": M.loss_criterion}
    for k, m in mods.items():
        synth.load_synth(m, SEEDS[k])
    orc = OracleStage3({k: {kk: vv.detach().clone() for kk, vv in m.state_dict().items()} for k, m in mods.items()})
    r = orc.train_step(_host(synth.stage4_batch(seed, B)), used=used)
    out = {"meta.seed": np.int64(seed), "meta.B": np.int64(B), "meta.used": np.array(used, np.int64),
           "final_output": r["final_output"].numpy().astype(np.float32), "losses": _losses(r)}
    for n in ("accu", "inpaint", "bg", "refine", "D", "face"):
        _grads(out, "", n, orc.sd[n])
    return out


CASES = {
    "fwd_s300_b1": lambda: case_forward(300, 1),
    "s300_b1": lambda: case_stage4(300, 1, steps=2),
    "s322_b2": lambda: case_stage4(322, 2),
    "s328_b8": lambda: case_stage4(328, 8),
    "s330_u2_p2": lambda: case_stage4(330, 1, (2,), 2),
    "s330_u30_p3": lambda: case_stage4(330, 1, (3, 0), 3),
    "s330_u123_p2": lambda: case_stage4(330, 1, (1, 2, 3), 2),
    "s330_u0123_p0": lambda: case_stage4(330, 1),
    "ranks2_s340": lambda: case_ranks(340, 2, (0, 1, 2, 3), 1, -1),
    "ranks2_s340_drop1": lambda: case_ranks(340, 2, (0, 1, 2, 3), 1, 1),
    "clip_s400": lambda: case_clip(400, 2, 3),
    "clip30_s401": lambda: case_clip30(401, 1, 30),
    "fwd512_s500_b1": lambda: case_forward(500, 1, S=512),
    "s501_b1_512": lambda: case_stage4(501, 1, S=512),
    "stage12_u0123": lambda: case_stage12((0, 1, 2, 3)),
    "stage12_u20": lambda: case_stage12((2, 0)),
    "stage3_s630_b2": lambda: case_stage3(),
}


def main(argv):
    names = argv or list(CASES)
    torch.set_num_threads(os.cpu_count())
    for n in names:
        t0 = time.time()
        out = CASES[n]()
        out["meta.torch"] = np.array(torch.__version__)
        out["meta.threads"] = np.int64(torch.get_num_threads())
        path = os.path.join(GOLD, "step_%s.npz" % n)
        np.savez_compressed(path, **out)
        INDEX.save()
        print("%-22s %6.1f s  %7.2f MB" % (n, time.time() - t0, os.path.getsize(path) / 1e6), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
