"""Parity of the stage-4 train step in the configurations that are actually benchmarked and trained
(VERDICT r1 "next round" item 1): batches above one sample (train-mode BatchNorm statistics, the 3x accumulating
discriminator loop and the face crops then see several samples), the bf16 matrix-core mode's gradients, reference
subsets `used` with a propagation source other than reference 0 (train/4...py:249-298), and two data-parallel ranks
against the oracle run on two chunks with averaged gradients (train/4...py:123-162)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests._step_util import (LOSSES, TRAINABLE, bn_buffers_err, build, check_losses, host, module_grad_rel, ref_keyed,
                              rel_l2)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mem_available_gb():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


# fp32 bars: frame <= 1e-3 L-inf (north star); losses 2e-3 relative; per-module gradient rel-L2 5e-3 (the L1 terms'
# sign() flips where |a-b| ~ 1e-6; measured 2e-6 .. 2e-3); BatchNorm running statistics 1e-4 relative.
@pytest.mark.parametrize("B", [2, 8])
def test_train_step_parity_batched(B):
    if B == 8 and _mem_available_gb() < 56:
        pytest.skip("the CPU oracle needs ~40 GB of host memory at B=8")
    M, tr, orc, batch, dbatch, mods = build(B, seed=320 + B)
    out = tr.train_step(dbatch)
    ref = orc.train_step(host(batch))
    err = (out["final_output"].cpu() - ref["final_output"]).abs().max().item()
    print("B=%d frame max|diff| %.3e" % (B, err))
    assert err <= 1e-3
    check_losses(out, ref, 2e-3, "B=%d" % B)
    for n in TRAINABLE:
        rel = module_grad_rel(mods[n], orc.sd[n])
        print("B=%d grad rel-L2 %-8s %.3e" % (B, n, rel))
        assert rel <= 5e-3, (n, rel)
    for n in ("flow", "D", "face"):
        e = bn_buffers_err(mods[n], orc.sd[n])
        print("B=%d BN running stats %-5s %.3e" % (B, n, e))
        assert e <= 1e-4, (n, e)
    for n in TRAINABLE:
        assert tr.flat[n].step_count == (3 if n == "D" else 1)


# `used`: 1-4 references in the (unsorted) order np.random.choice drew them, unused masks zeroed, the propagation
# source one of the used references with ITS OWN SMPL pose (train/4...py:249-298).
# Gradient bars at B=1: the loss is a sum of L1 terms over (Leaky)ReLU networks, so a 1e-6 forward difference flips
# sign()/slope decisions and the gradient difference grows module by module going upstream (refine -> inpaint -> accu);
# with ONE sample nothing averages it out.  Measured on this batch (seed 330), relative L2 per module:
#   CPU oracle in fp32 vs the same oracle in fp64 (the comparison's own floor): accu 2.2e-3, inpaint 1.9e-3, refine 9e-4;
#   GPU fp32 vs CPU oracle fp32: accu 6.3e-3 (all four references) / 8.0e-3 / 1.2e-2, inpaint 5.3e-3 / 6.5e-3,
#   uniform over every layer of a module (scratch/diag_subsets.py), i.e. inherited from the incoming gradient, not a
#   layer of its own.  At B=2 / B=8 the same quantities are 2.5e-3 / 2.4e-3 (test_train_step_parity_batched, bar 5e-3).
SUBSET_GRAD_BARS = {"accu": 3e-2, "inpaint": 2e-2, "refine": 1e-2, "flow": 5e-3, "D": 5e-3, "face": 5e-3}
@pytest.mark.parametrize("used,prosrc", [((2,), 2), ((3, 0), 3), ((1, 2, 3), 2)])
def test_train_step_parity_reference_subsets(used, prosrc):
    M, tr, orc, batch, dbatch, mods = build(1, seed=330)
    out = tr.train_step(dbatch, used=used, prosrc=prosrc)
    ref = orc.train_step(host(batch), used=used, prosrc=prosrc)
    err = (out["final_output"].cpu() - ref["final_output"]).abs().max().item()
    print("used=%s prosrc=%d frame max|diff| %.3e" % (used, prosrc, err))
    assert err <= 1e-3
    check_losses(out, ref, 2e-3, "used=%s" % (used,))
    for n in TRAINABLE:
        rel = module_grad_rel(mods[n], orc.sd[n])
        print("used=%s grad rel-L2 %-8s %.3e (bar %.0e)" % (used, n, rel, SUBSET_GRAD_BARS[n]))
        assert rel <= SUBSET_GRAD_BARS[n], (n, rel)
    # the source pose matters: the same step with reference 0's pose must give another warped frame
    if prosrc != 0:
        from jafpro_amd.step import generator_forward
        with torch.no_grad():
            a = generator_forward(M, dbatch, used, prosrc)["tsf_image"]
            alt = dict(dbatch)
            alt["src_verts_refs"] = dbatch["src_verts_refs"][:, [0, 0, 0, 0]].contiguous()
            b = generator_forward(M, alt, used, prosrc)["tsf_image"]
        assert (a - b).abs().max().item() > 1e-3


# bf16 mode (BASELINE configs[2] arithmetic, the bench default): operands of every convolution are rounded to 8
# significant bits, forward AND backward, and every L1 term's sign(a - b) flips wherever the 1e-2 forward perturbation
# exceeds |a - b|, so gradients agree with the fp32 oracle to 5-16 % in relative L2, not to 1e-3.  Bars = 2x what was
# measured on MI355X at B=2 (accu 0.155, inpaint 0.158, refine 0.079, flow 0.054, D 0.082, face 0.122); a wrong
# dgrad/wgrad kernel, a stale packed weight image or a dropped term shows up as O(1).
BF16_GRAD_BARS = {"accu": 0.30, "inpaint": 0.30, "refine": 0.16, "flow": 0.12, "D": 0.16, "face": 0.25}


def test_train_step_bf16_gradients():
    from jafpro_amd import ops
    M, tr, orc, batch, dbatch, mods = build(2, seed=322)
    prev = ops.set_precision("bf16")
    try:
        out = tr.train_step(dbatch)
    finally:
        ops.set_precision(prev)
    ref = orc.train_step(host(batch))
    check_losses(out, ref, 2e-2, "bf16")
    err = (out["final_output"].cpu() - ref["final_output"]).abs().max().item()
    print("bf16 B=2 frame max|diff| %.3e rel-L2 %.3e" % (err, rel_l2(out["final_output"].cpu(), ref["final_output"])))
    assert err <= 1e-1
    for n in TRAINABLE:
        rel = module_grad_rel(mods[n], orc.sd[n])
        print("bf16 grad rel-L2 %-8s %.3e (bar %.2f)" % (n, rel, BF16_GRAD_BARS[n]))
        assert rel <= BF16_GRAD_BARS[n], (n, rel)
    for n in ("flow", "D", "face"):
        assert bn_buffers_err(mods[n], orc.sd[n]) <= 5e-2, n


def test_train_step_bf16x3_is_parity_grade():
    """The split-bf16 mode (three bf16 MFMAs per product, forward, data AND weight gradients on the matrix cores) held to
    the fp32 bars of the full step: frame <= 1e-3 L-inf, losses 2e-3, per-module gradients at the B=1 bars above."""
    from jafpro_amd import ops
    M, tr, orc, batch, dbatch, mods = build(1, seed=330)
    prev = ops.set_precision("bf16x3")
    try:
        out = tr.train_step(dbatch)
    finally:
        ops.set_precision(prev)
    ref = orc.train_step(host(batch))
    err = (out["final_output"].cpu() - ref["final_output"]).abs().max().item()
    print("bf16x3 frame max|diff| %.3e" % err)
    assert err <= 1e-3
    check_losses(out, ref, 2e-3, "bf16x3")
    for n in TRAINABLE:
        rel = module_grad_rel(mods[n], orc.sd[n])
        print("bf16x3 grad rel-L2 %-8s %.3e (bar %.0e)" % (n, rel, SUBSET_GRAD_BARS[n]))
        assert rel <= SUBSET_GRAD_BARS[n], (n, rel)


def test_bf16_second_step_uses_refreshed_weight_images():
    """Two bf16 steps; the packed weight images were re-made IN PLACE on a side stream after each Adam
    (ops.refresh_packed_weights).  A forward that uses those cached images must equal, bit for bit, a forward that
    packs every image from scratch from the current weights -- a stale or half-written image would differ."""
    from jafpro_amd import ops
    from jafpro_amd.step import generator_forward
    M, tr, orc, batch, dbatch, mods = build(1)
    prev = ops.set_precision("bf16")
    try:
        tr.train_step(dbatch, next_batch=dbatch)
        out = tr.train_step(dbatch)
        assert all(torch.isfinite(out[k]).all() for k in LOSSES)
        with torch.no_grad():
            g_cached = generator_forward(M, dbatch, (0, 1, 2, 3), 0)        # refreshed images, straight from the cache
            ops.invalidate_packed_weights()
            g_fresh = generator_forward(M, dbatch, (0, 1, 2, 3), 0)
    finally:
        ops.set_precision(prev)
    for k in ("accu", "inpaint", "refine_output", "fusion_output", "final_output"):
        assert torch.equal(g_cached[k], g_fresh[k]), k
    # and the weights did move: the same forward before the two steps gave another frame
    M0, _, _, _, _, _ = build(1)
    prev = ops.set_precision("bf16")
    try:
        with torch.no_grad():
            g0 = generator_forward(M0, dbatch, (0, 1, 2, 3), 0)
    finally:
        ops.set_precision(prev)
    assert not torch.equal(g0["final_output"], g_fresh["final_output"])


def _run_ranks(tmp_path, world, precision, seed, used, prosrc, drop_face_rank=-1):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    procs, outs = [], []
    for r in range(world):
        outp = str(tmp_path / ("rank%d.pt" % r))
        outs.append(outp)
        cmd = [sys.executable, os.path.join(ROOT, "tests", "_rank_worker.py"), str(r), str(world), str(port), outp, precision,
               str(seed), ",".join(str(u) for u in used), str(prosrc)] + ([str(drop_face_rank)] if drop_face_rank >= 0 else [])
        procs.append(subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(o, weights_only=False) for o in outs]


@pytest.mark.parametrize("drop_face_rank", [-1, 1])
def test_two_rank_trainer_vs_chunked_oracle(tmp_path, drop_face_rank):
    """N ranks == the single-process restatement with the batch split in N chunks for the BatchNorm
    statistics and the gradients averaged.  Two real trainer processes (B=1 each, gradient messages started from inside
    the backward pass) against oracle.train_step_ranks on the same two shards: averaged gradients of all six modules,
    post-Adam parameters, rank-local BatchNorm buffers, per-rank losses and frames.  drop_face_rank=1: rank 1 holds no
    valid face box -- it must still join every exchange, and the face terms must be the mean over the faces that exist."""
    from jafpro_amd import synth
    from jafpro_amd.dist import shard_batch
    from oracle.step_oracle import OracleStage4
    from tests._step_util import build_models
    used, prosrc, seed = (0, 1, 2, 3), 1, 340
    res = _run_ranks(tmp_path, 2, "f32", seed, used, prosrc, drop_face_rank)
    _, _, sds, fidx = build_models()
    orc = OracleStage4(sds, fidx)
    full = synth.stage4_batch(seed, 2)
    if drop_face_rank >= 0:
        full["face_bbox"][drop_face_rank] = (96, 96, 32, 96)
    before = {n: {k: v.detach().clone() for k, v in orc.sd[n].items() if v.requires_grad} for n in TRAINABLE}
    refs = orc.train_step_ranks([host(shard_batch(full, r, 2)) for r in range(2)], used, prosrc)
    views = orc._rank_views(2)
    for r in range(2):
        assert res[r]["overlap_order"] == ["flow", "refine", "inpaint", "accu"]
        err = (res[r]["final_output"] - refs[r]["final_output"]).abs().max().item()
        print("rank %d frame max|diff| %.3e" % (r, err))
        assert err <= 1e-3
        for k in LOSSES:
            a, b = res[r]["losses"][k], float(refs[r][k].reshape(-1)[0])
            assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (r, k, a, b)
        for n in ("flow", "D", "face"):
            for k, v in res[r]["buffers"][n].items():
                o = views[r][n][k]
                if k.endswith("num_batches_tracked"):
                    if not (n == "face" and r == drop_face_rank):
                        assert int(v) == int(o), (r, n, k, int(v), int(o))
                else:
                    assert (v - o).abs().max().item() <= 1e-4 * max(1e-6, o.abs().max().item()) + 1e-7, (r, n, k)
    for n in TRAINABLE:
        num = den = pnum = pden = 0.0
        gsq = cnt = 0.0
        for k, p in orc.sd[n].items():
            if p.requires_grad:
                gsq += float((p.grad.double() ** 2).sum()); cnt += p.numel()
        rms = (gsq / cnt) ** 0.5
        for k, p in orc.sd[n].items():
            if not p.requires_grad:
                continue
            g0, g1 = res[0]["grads"][n][k], res[1]["grads"][n][k]
            if n in ("D", "face"):
                # what is left in the discriminators' buffers = the all-reduced gradients of their own updates + the
                # never-used deposit of the generator's backward pass (F10), which stays rank-local: the ranks differ by
                # that deposit and their MEAN is the oracle's buffer
                g0 = (g0.double() + g1.double()) / 2
            else:
                assert torch.equal(g0, g1), (n, k)                  # both ranks hold the same averaged gradient
            d = g0.double() - p.grad.double()
            num += float((d * d).sum()); den += float((p.grad.double() ** 2).sum())
            assert torch.equal(res[0]["params"][n][k], res[1]["params"][n][k]), (n, k)
            # Adam moves an element by ~lr * sign(g) on its first step(s): where the gradient is not small against the
            # module's RMS gradient its sign is certain and the updates must agree; elsewhere a 1e-3 gradient error may flip it
            sure = p.grad.abs() > 0.05 * rms
            du_ref = (p.detach().double() - before[n][k].double())[sure]
            du = (res[0]["params"][n][k].double() - before[n][k].double())[sure]
            pnum += float(((du - du_ref) ** 2).sum()); pden += float((du_ref ** 2).sum())
        rel, prel = (num / max(den, 1e-300)) ** 0.5, (pnum / max(pden, 1e-300)) ** 0.5
        print("2 ranks: grad rel-L2 %-8s %.3e   Adam update rel-L2 (|g| > 5%% of rms) %.3e" % (n, rel, prel))
        assert rel <= 5e-3, (n, rel)
        assert prel <= 1e-2, (n, prel)
