"""stage-1..3 train steps on the HIP modules against the CPU restatements of train/1-3*.py
(oracle/stage_oracle.py), BASELINE configs[0] against the golden made from the reference module, and checkpoint
files through a GPU trainer."""
import numpy as np
import pytest
import torch

from tests._step_util import SEEDS, check_losses_golden, golden_grad_rel, golden_step, rel_l2

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_stage1_config1_step_golden(golden_dir):
    """BASELINE configs[0] (stage-1 text_accu_LSTM, one 4-frame clip, B=1): loss, atlas, gradients and the Adam(1e-4)
    update of Stage1Trainer vs the golden recorded from the reference's Accumulate_LSTM (train/1...py:140-176)."""
    import os
    from jafpro_amd import synth
    from jafpro_amd.networks import Accumulate_LSTM
    from jafpro_amd.stages import Stage1Trainer
    from tests._step_util import ref_keyed
    st = dict(np.load(os.path.join(golden_dir, "stage1_b1_t4_step.npz")))
    m = synth.load_synth(Accumulate_LSTM(), 111).cuda()
    before = {k: v.clone() for k, v in ref_keyed(m, "data").items()}
    tr = Stage1Trainer(m)
    out = tr.train_step({k: T(v) for k, v in synth.stage1_batch(611, 1).items()}, (0, 1, 2, 3))
    assert abs(float(out["total_loss"]) - float(st["loss"][0])) <= 1e-5
    flat = out["output_texture"].reshape(-1).cpu()
    assert (flat[torch.from_numpy(st["atlas.idx"])] - torch.from_numpy(st["atlas.samples"])).abs().max().item() <= 1e-3
    g, after = ref_keyed(m), ref_keyed(m, "data")
    for k in [k[5:] for k in st if k.startswith("grad.")]:
        ref = torch.from_numpy(st["grad." + k])
        assert rel_l2(g[k].cpu(), ref) <= 2e-3, (k, rel_l2(g[k].cpu(), ref))
        d = (after[k] - before[k]).cpu()
        dref = torch.from_numpy(st["delta." + k])
        # first Adam step = -lr * g/(|g|+eps): elements whose gradient is not ~0 move by exactly -lr*sign(g)
        big = ref.abs() > 1e-6
        assert (d[big] - dref[big]).abs().max().item() <= 2e-6, k
    fam = {}
    for k, v in g.items():
        f = ".".join(k.split(".")[2:])
        fam[f] = fam.get(f, 0.0) + float((v.double() ** 2).sum())
    for f, v in zip(st["gradsq.families"], st["gradsq.values"]):
        assert abs(fam[str(f)] - float(v)) <= 5e-3 * float(v), (f, fam[str(f)], float(v))
    assert tr.flat["accu"].step_count == 1


def _strided_err(t, gold, key, n=65536):
    from oracle import step_digest as SD
    flat = t.reshape(-1)
    got = flat[::SD.stride_for(flat.numel(), n)].cpu()
    sq = float((flat.double() ** 2).sum())
    assert abs(sq - float(gold[key + ".sq"])) <= 1e-3 * float(gold[key + ".sq"]), (key, sq)
    return (got - torch.from_numpy(gold[key + ".strided"])).abs().max().item()


@pytest.mark.parametrize("used", [(0, 1, 2, 3), (2, 0)])
def test_stage1_and_stage2_steps_vs_oracle(used):
    """Stage-1 and stage-2 steps vs oracle/stage_oracle.py (fixtures stage12_u*.npz made by oracle/make_step_golden.py)."""
    from jafpro_amd import synth
    from jafpro_amd.networks import Accumulate_LSTM, Accumulate_LSTM_no_loss, UNet_inpainter
    from jafpro_amd.stages import Stage1Trainer, Stage2Trainer
    gold = golden_step("stage12_u" + "".join(str(u) for u in used))
    db = {k: T(v) for k, v in synth.stage1_batch(620, 1).items()}
    m1 = synth.load_synth(Accumulate_LSTM(), 121)
    t1 = Stage1Trainer(m1.cuda())
    out = t1.train_step(db, used)
    assert abs(float(out["total_loss"]) - float(gold["s1.total_loss"])) <= 1e-5
    assert _strided_err(out["output_texture"], gold, "s1.output_texture") <= 1e-3
    r, worst, _ = golden_grad_rel(m1, gold, "accu", prefix="s1.")
    print("stage 1 used=%s grad rel-L2 %.3e (worst tensor-norm deviation %.3e)" % (used, r, worst))
    assert r <= 5e-3 and worst <= 5e-2
    accu, inp = synth.load_synth(Accumulate_LSTM_no_loss(), 122), synth.load_synth(UNet_inpainter(), 123)
    t2 = Stage2Trainer(accu.cuda(), inp.cuda())
    out = t2.train_step(db, used)
    assert abs(float(out["total_loss"]) - float(gold["s2.total_loss"])) <= 1e-4 * max(1.0, float(gold["s2.total_loss"]))
    assert _strided_err(out["inpaint"], gold, "s2.inpaint") <= 1e-3
    for n, mod in (("accu", accu), ("inpaint", inp)):
        r, worst, _ = golden_grad_rel(mod, gold, n, prefix="s2.")
        print("stage 2 used=%s grad rel-L2 %-8s %.3e (worst tensor-norm deviation %.3e)" % (used, n, r, worst))
        assert r <= 5e-3 and worst <= 5e-2, (n, r, worst)
        assert t2.flat[n].step_count == 1


def test_stage3_step_vs_oracle():
    """train/3.inpaint_global_convLSTM_FGAN.py:193-382: trainable background CRN, three accumulating face-D and image-D
    updates, face GAN term through the (non-detached) crop; vs OracleStage3 (fixture stage3_s630_b2)."""
    from jafpro_amd import synth
    from jafpro_amd.stages import Stage3Models, Stage3Trainer
    from jafpro_amd.step import _to_dev
    M = Stage3Models()
    mods = {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
            "D": M.discriminator, "face": M.F_Discriminator, "vgg": M.loss_criterion}
    for k, m in mods.items():
        synth.load_synth(m, SEEDS[k])
    M = M.cuda()
    dbatch = _to_dev(synth.stage4_batch(630, 2), "cuda")
    gold = golden_step("stage3_s630_b2")
    tr = Stage3Trainer(M)
    out = tr.train_step(dbatch, used=(1, 3, 0))
    assert (out["final_output"].cpu() - torch.from_numpy(gold["final_output"])).abs().max().item() <= 1e-3
    check_losses_golden(out, gold["losses"], 2e-3, "stage3")
    for n in ("accu", "inpaint", "bg", "refine", "D", "face"):
        r, worst, _ = golden_grad_rel(mods[n], gold, n)
        print("stage 3 grad rel-L2 %-8s %.3e (worst tensor-norm deviation %.3e)" % (n, r, worst))
        assert r <= 5e-3 and worst <= 5e-2, (n, r, worst)
        assert tr.flat[n].step_count == (3 if n in ("D", "face") else 1)


def test_checkpoints_through_a_gpu_trainer(tmp_path):
    """Save the seven stage-4 files from a trainer that has taken a step (parameters live in flat buffers), load them
    into a fresh model set: identical state_dicts and a bit-identical forward; loading INTO a trainer keeps its flat
    buffers and drops stale packed weight images."""
    import os
    from jafpro_amd import ops, stages
    from jafpro_amd.step import Stage4Trainer, generator_forward
    from tests._step_util import build
    M, tr, _, _, dbatch, _ = build(1)
    tr.train_step(dbatch)
    with torch.no_grad():       # (train-mode BatchNorm: this forward moves the running statistics, so it comes before the save)
        a = generator_forward(M, dbatch, (0, 1, 2, 3), 0)["final_output"]
    paths = stages.save_checkpoints(str(tmp_path), 12, stages.stage4_modules(M))
    assert sorted(os.path.basename(p) for p in paths.values()) == sorted(
        "%s_iter_12.pth" % p for p in ("Accu", "inpaint", "bg", "refine", "D", "FD", "pro"))
    M2, tr2, _, _, _, _ = build(1)
    with torch.no_grad():
        before = generator_forward(M2, dbatch, (0, 1, 2, 3), 0)["final_output"]
    ptr = tr2.flat["refine"].flat.data_ptr()
    for name, m in stages.stage4_modules(M2).items():
        stages.load_checkpoint(m, paths[name])
    assert tr2.flat["refine"].flat.data_ptr() == ptr and M2.refine_model.out_conv.weight.data_ptr() >= ptr
    for (k1, v1), (k2, v2) in zip(M.state_dict().items(), M2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    with torch.no_grad():
        b = generator_forward(M2, dbatch, (0, 1, 2, 3), 0)["final_output"]
    assert not torch.equal(before, a) and torch.equal(a, b)
