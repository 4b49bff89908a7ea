"""Shared helpers of the stage-4 step parity tests (GPU trainer vs oracle/step_oracle.py)."""
import numpy as np
import torch

SEEDS = {"accu": 201, "inpaint": 202, "bg": 203, "refine": 204, "flow": 205, "D": 206, "face": 207, "vgg": 208}
TRAINABLE = ("accu", "inpaint", "refine", "flow", "D", "face")
LOSSES = ("total_loss", "vgg_l1", "errD", "errG", "F_errD", "F_errG")


def build_models(image_size=256):
    from jafpro_amd import synth
    from jafpro_amd.step import Stage4Models
    _, fidx = synth.body_mesh()
    M = Stage4Models(fidx, image_size=image_size)
    mods = {"accu": M.Accu_model, "inpaint": M.inpaint_model, "bg": M.bg_model, "refine": M.refine_model,
            "flow": M.propagater, "D": M.discriminator, "face": M.F_Discriminator, "vgg": M.loss_criterion}
    for k, m in mods.items():
        synth.load_synth(m, SEEDS[k])
    sds = {k: {kk: vv.detach().clone() for kk, vv in m.state_dict().items()} for k, m in mods.items()}
    return M, mods, sds, fidx


def build(B, seed=300, reducer=None):
    """-> (models on the GPU, trainer, oracle, host batch, device batch, {name: module})."""
    from jafpro_amd import synth
    from jafpro_amd.step import Stage4Trainer, _to_dev
    from oracle.step_oracle import OracleStage4
    M, mods, sds, fidx = build_models()
    M = M.cuda()
    batch = synth.stage4_batch(seed, B)
    return M, Stage4Trainer(M, reducer=reducer), OracleStage4(sds, fidx), batch, _to_dev(batch, "cuda"), mods


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
