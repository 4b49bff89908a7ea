"""Shared helpers of the stage-4 step parity tests (GPU trainer vs oracle/step_oracle.py)."""
import numpy as np
import torch

SEEDS = {"accu": 201, "inpaint": 202, "bg": 203, "refine": 204, "flow": 205, "D": 206, "face": 207, "vgg": 208}
TRAINABLE = ("accu", "inpaint", "refine", "flow", "D", "face")
LOSSES = ("total_loss", "vgg_l1", "errD", "errG", "F_errD", "F_errG")


def stage4_modules(M):
    return {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
            "flow": M.propagater, "D": M.discriminator, "face": M.F_Discriminator, "vgg": M.loss_criterion}


def build_models(image_size=256):
    """CPU model set with the portable synthetic weights of SEEDS (+ its reference-keyed state_dicts)."""
    from jafpro_amd import synth
    from jafpro_amd.step import Stage4Models
    _, fidx = synth.body_mesh()
    M = Stage4Models(fidx, image_size=image_size)
    mods = stage4_modules(M)
    for k, m in mods.items():
        synth.load_synth(m, SEEDS[k])
    sds = {k: {kk: vv.detach().clone() for kk, vv in m.state_dict().items()} for k, m in mods.items()}
    return M, mods, sds, fidx


_PRISTINE = {}


def gpu_models(image_size=256):
    """A fresh model set on the GPU holding the SEEDS weights.  Building one on the host costs ~13 s (default
    initialisation of 96 M parameters, the synthetic fill, the 24-part regrouping), so ONE pristine copy per process
    stays on the device and every test gets a device-to-device deep copy of it."""
    import copy
    if 256 not in _PRISTINE:
        M, _, _, _ = build_models(256)
        _PRISTINE[256] = M.cuda()
    M = copy.deepcopy(_PRISTINE[256])
    if image_size != 256:                   # the renderer's raster size is the only thing the frame size changes
        M.image_size = M.flow_calculator.render.image_size = image_size
    return M, stage4_modules(M)


def build(B, seed=300, reducer=None, oracle=False):
    """-> (models on the GPU, trainer, oracle or None, host batch, device batch, {name: module}).
    The CPU oracle is only built on request: the parity tests compare with tests/golden/step_*.npz."""
    from jafpro_amd import synth
    from jafpro_amd.step import Stage4Trainer, _to_dev
    M, mods = gpu_models()
    orc = None
    if oracle:
        from oracle.step_oracle import OracleStage4
        _, fidx = synth.body_mesh()
        orc = OracleStage4({k: {kk: vv.detach().cpu().clone() for kk, vv in m.state_dict().items()} for k, m in mods.items()}, fidx)
    batch = synth.stage4_batch(seed, B)
    return M, Stage4Trainer(M, reducer=reducer), orc, batch, _to_dev(batch, "cuda"), mods


def host(batch):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}


def rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def ref_keyed(module, what="grad"):
    """{reference state_dict key: gradient (or value) of that parameter} for a jafpro_amd module; the 24-part
    networks store one grouped parameter per layer (networks._GroupedStateDict) and are cut back per part."""
    out = {}
    km = getattr(module, "_key_map", None) or {}
    for name, p in module.named_parameters():
        t = p.grad if what == "grad" else p.data
        if t is None:
            continue
        t = t.detach()
        if name in km:
            per = t.shape[0] // 24
            for q in range(24):
                out[km[name].format(p=q)] = t[q * per:(q + 1) * per]
        else:
            out[name] = t
    return out


def module_grad_rel(module, osd):
    """relative L2 distance, over ALL trainable parameters of a module, between the gradients left in the GPU
    module's buffers and the oracle's .grad."""
    g = ref_keyed(module)
    num = den = 0.0
    for k, p in osd.items():
        if not p.requires_grad:
            continue
        d = g[k].cpu().double() - p.grad.double()
        num += float((d * d).sum())
        den += float((p.grad.double() ** 2).sum())
    return (num / max(den, 1e-300)) ** 0.5


def bn_buffers_err(module, osd):
    """max relative error over the BatchNorm running statistics (+ exact num_batches_tracked)."""
    worst = 0.0
    for k, v in module.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(osd[k]), (k, int(v), int(osd[k]))
        elif "running_" in k:
            e = (v.cpu() - osd[k]).abs().max().item() / max(1e-6, osd[k].abs().max().item())
            worst = max(worst, e)
    return worst


def check_losses(out, ref, tol, tag=""):
    for k in LOSSES:
        a, b = float(out[k].reshape(-1)[0]), float(ref[k].reshape(-1)[0])
        print("%s %-10s gpu %.6f cpu %.6f" % (tag, k, a, b))
        assert np.isfinite(a) and abs(a - b) <= tol * max(1.0, abs(b)), (tag, k, a, b)


# ------------------------------------------------------------------------------------------------
# comparisons against tests/golden/step_*.npz (made by oracle/make_step_golden.py in the build container)
# ------------------------------------------------------------------------------------------------
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_INDEX = {}


def golden_step(case):
    return dict(np.load(os.path.join(GOLD, "step_%s.npz" % case)))


def step_index():
    if not _INDEX:
        _INDEX.update(np.load(os.path.join(GOLD, "step_index.npz")))
    return _INDEX


def flat_in_reference_order(module, name, what="grad"):
    """The module's gradients (or parameters) as ONE device vector in the reference's state_dict order -- the order of
    the fixture's sample positions."""
    keys = [str(k) for k in step_index()["keys." + name]]
    t = ref_keyed(module, what)
    return torch.cat([t[k].reshape(-1) for k in keys])


def golden_grad_rel(module, gold, name, prefix="", index_name=None, flat=None):
    """(relative L2 distance of the module's gradient from the oracle's on the fixture's sample of positions,
    worst relative deviation of a parameter tensor's gradient NORM over the tensors that carry >= 1e-2 of the module's
    gradient norm, that tensor's reference key)."""
    ix = step_index()
    iname = index_name or name
    if flat is None:
        flat = flat_in_reference_order(module, iname)
    idx = torch.from_numpy(ix["idx." + iname]).to(flat.device)
    ref = torch.from_numpy(gold["%sg.%s.val" % (prefix, name)]).to(flat.device).double()
    d = flat[idx].double() - ref
    rel = float((d * d).sum().sqrt() / ref.pow(2).sum().sqrt().clamp_min(1e-300))
    numel = torch.from_numpy(ix["numel." + iname])
    assert int(numel.sum()) == flat.numel(), (name, int(numel.sum()), flat.numel())
    ends = torch.cumsum(numel, 0).to(flat.device)
    cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=flat.device), torch.cumsum(flat.double() ** 2, 0)])
    sq = (cs[ends] - cs[ends - numel.to(flat.device)]).cpu()
    sq_ref = torch.from_numpy(gold["%sg.%s.sq" % (prefix, name)])
    n, n_ref = sq.clamp_min(0).sqrt(), sq_ref.sqrt()
    heavy = n_ref >= 1e-2 * float(sq_ref.sum().sqrt())
    dev = ((n - n_ref).abs() / n_ref.clamp_min(1e-300)) * heavy
    at = int(dev.argmax())
    return rel, float(dev[at]), str(ix["keys." + iname][at])


def golden_bn_err(module, gold, name, prefix=""):
    """max relative error of the BatchNorm running statistics vs the fixture (num_batches_tracked exact)."""
    worst = 0.0
    for k, v in module.state_dict().items():
        key = "%sbn.%s.%s" % (prefix, name, k)
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(gold[key]), (k, int(v), int(gold[key]))
        elif "running_" in k:
            o = torch.from_numpy(gold[key])
            worst = max(worst, (v.cpu() - o).abs().max().item() / max(1e-6, o.abs().max().item()))
    return worst


def check_losses_golden(out, ref_losses, tol, tag=""):
    for k, b in zip(LOSSES, ref_losses):
        a, b = float(out[k].reshape(-1)[0]) if not isinstance(out[k], float) else out[k], float(b)
        print("%s %-10s gpu %.6f cpu %.6f" % (tag, k, a, b))
        assert np.isfinite(a) and abs(a - b) <= tol * max(1.0, abs(b)), (tag, k, a, b)


def check_step_golden(case, out, mods, frame_tol=1e-3, loss_tol=2e-3, grad_bars=5e-3, bn_tol=1e-4, norm_factor=10.0, tag=None):
    """One train step's results against tests/golden/step_<case>.npz: whole frame (L-inf), six losses, per-module
    gradient, BatchNorm buffers.  The gradient is held to `grad_bars` in relative L2 on the fixture's sample of the flat
    gradient vector; on top of that EVERY parameter tensor that carries at least 1 % of its module's gradient norm must
    have that norm within norm_factor x bar of the oracle's (a dropped or doubled term in one layer is O(1); the sign-flip
    noise of an L1 loss over (Leaky)ReLU networks concentrates in single small tensors at several times the module's
    average, measured up to 2.4e-2 at B <= 2 in fp32)."""
    gold = golden_step(case)
    tag = tag or case
    err = (out["final_output"].cpu() - torch.from_numpy(gold["final_output"])).abs().max().item()
    print("%s frame max|diff| %.3e" % (tag, err))
    assert err <= frame_tol, (tag, err)
    check_losses_golden(out, gold["losses"], loss_tol, tag)
    for n in TRAINABLE:
        bar = grad_bars[n] if isinstance(grad_bars, dict) else grad_bars
        rel, worst, where = golden_grad_rel(mods[n], gold, n)
        print("%s grad rel-L2 %-8s %.3e (bar %.0e)   worst tensor-norm deviation %.3e (%s)" % (tag, n, rel, bar, worst, where))
        assert rel <= bar, (tag, n, rel)
        assert worst <= norm_factor * bar, (tag, n, worst, where)
    if bn_tol is not None:
        for n in ("flow", "D", "face"):
            e = golden_bn_err(mods[n], gold, n)
            print("%s BN running stats %-5s %.3e" % (tag, n, e))
            assert e <= bn_tol, (tag, n, e)
    return gold
