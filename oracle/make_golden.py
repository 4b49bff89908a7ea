"""GOLDEN-VECTOR GENERATOR (survey container only; needs /root/reference, never runs on the GPU box).

Imports the reference's own nn.Modules through the shim of SURVEY.md Appendix C, loads the
portable synthetic weights of jafpro_amd/synth.py, runs them on seeded inputs and
  (1) asserts that oracle/torch_oracle.py reproduces every output (pins the oracle), and
  (2) writes inputs-by-seed + expected outputs (full small tensors, or digest + 4096 strided
      samples of large ones) to tests/golden/*.npz.
Only data is written; no reference source travels.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

# ---- shim -------------------------------------------------------------------
tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
tvm.vgg19 = tvm.vgg16 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
tv.models = tvm
sys.modules.update({"torchvision": tv, "torchvision.models": tvm})
torch.Tensor.cuda = lambda self, *a, **k: self          # F11

import src.networks as RN                                # noqa: E402
from src.convLSTM import ConvLSTM as RConvLSTM           # noqa: E402
from src.crn_model import CRN_smaller as RCRN            # noqa: E402
from src.flow_net import Propagation3DFlowNet as RPro    # noqa: E402

from jafpro_amd import synth                             # noqa: E402
from oracle import torch_oracle as O                     # noqa: E402
from oracle import raster_oracle                         # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_grad_enabled(True)


def digest(t: torch.Tensor, n=4096):
    a = t.detach().double().reshape(-1)
    idx = np.linspace(0, a.numel() - 1, min(n, a.numel())).astype(np.int64)
    return {"shape": np.array(t.shape, np.int64), "sum": np.float64(a.sum().item()),
            "sumabs": np.float64(a.abs().sum().item()), "sumsq": np.float64((a * a).sum().item()),
            "idx": idx, "samples": a[idx].float().numpy()}


def put(store, name, t, full=False):
    if full:
        store[name] = t.detach().float().numpy()
    else:
        for k, v in digest(t).items():
            store["%s.%s" % (name, k)] = v


def close(a, b, tol, what):
    err = (a.detach() - b.detach()).abs().max().item()
    print("  oracle vs reference %-28s max|diff| = %.3e" % (what, err))
    assert err <= tol, (what, err)
    return err


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def sd_of(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def grads_of(module, loss):
    module.zero_grad()
    loss.backward()
    return {k: p.grad.detach().clone() for k, p in module.named_parameters() if p.grad is not None}


report = {}

# ---- 1. ConvLSTM toy ------------------------------------------------------------------------------
def g_convlstm():
    st = {}
    m = synth.load_synth(RConvLSTM((7, 5), 4, [4], [(3, 3)], 1, batch_first=True, bias=True), 11)
    x = T(synth.uniform(11, "x", (2, 3, 4, 7, 5))).requires_grad_(True)
    out, last = m(x)
    proj = T(synth.uniform(11, "proj", tuple(last[0][0].shape)))
    loss = (last[0][0] * proj).sum() + 0.5 * (out * out).sum()
    g = grads_of(m, loss)
    sd = sd_of(m)
    hs, (h, c) = O.convlstm(sd["cell_list.0.conv.weight"], sd["cell_list.0.conv.bias"], [x[:, t] for t in range(3)])
    report["convlstm"] = close(torch.stack(hs, 1), out, 1e-6, "ConvLSTM out")
    close(c, last[0][1], 1e-6, "ConvLSTM c_T")
    put(st, "out", out, True); put(st, "h_T", last[0][0], True); put(st, "c_T", last[0][1], True)
    put(st, "dx", x.grad, True)
    put(st, "dw", g["cell_list.0.conv.weight"], True); put(st, "db", g["cell_list.0.conv.bias"], True)
    np.savez_compressed(os.path.join(GOLD, "convlstm_toy.npz"), **st)


# ---- 2. Accumulate_LSTM_no_loss, full size ----------------------------------------------------------
def accu_inputs(seed, B, Tn):
    atlas = synth.uniform(seed, "src_texture_im", (B, Tn, 3, 800, 1200))
    x_in = []
    for i in range(4):
        for j in range(6):
            x_in.append([T(atlas[:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200]) for t in range(Tn)])
    return x_in


def g_accumulate():
    st = {}
    m = synth.load_synth(RN.Accumulate_LSTM_no_loss(), 21)
    x_in = accu_inputs(21, 1, 2)
    outs = m(x_in)
    out = torch.cat(outs, 1)
    proj = T(synth.uniform(21, "proj", tuple(out.shape)))
    g = grads_of(m, (out * proj).sum())
    oo = torch.cat(O.accumulate_forward(sd_of(m), x_in), 1)
    report["accumulate"] = close(oo, out, 2e-5, "Accumulate_LSTM_no_loss")
    put(st, "out", out)
    for k in ["Downsampler_list.0.enc1.enconv.0.weight", "Downsampler_list.5.convLSTM1.cell_list.0.conv.weight",
              "Downsampler_list.23.convLSTM5.cell_list.0.conv.bias", "Upsampler_list.7.dec1.myconv.0.weight",
              "Upsampler_list.23.conv.bias", "Downsampler_list.11.enc8.enconv.0.weight"]:
        put(st, "grad." + k, g[k], True)
    np.savez_compressed(os.path.join(GOLD, "accumulate_b1_t2.npz"), **st)


# ---- 3. UNet_inpainter ------------------------------------------------------------------------------
def g_inpaint():
    st = {}
    m = synth.load_synth(RN.UNet_inpainter(), 31)
    tex = [T(synth.uniform(31, "tex%d" % p, (1, 3, 200, 200))).requires_grad_(True) for p in range(24)]
    outs = m(tex)
    out = torch.cat(outs, 1)
    proj = T(synth.uniform(31, "proj", tuple(out.shape)))
    g = grads_of(m, (out * proj).sum())
    oo = torch.cat(O.inpaint_forward(sd_of(m), [t.detach() for t in tex]), 1)
    report["inpaint"] = close(oo, out, 2e-5, "UNet_inpainter")
    put(st, "out", out)
    put(st, "dtex3", tex[3].grad)
    for k in ["Downsampler_list.2.enc_compress.enconv.0.weight", "Upsampler_list.0.dec1.myconv.0.weight",
              "Downsampler_list.9.enc1.enconv.0.bias", "Upsampler_list.20.conv.weight"]:
        put(st, "grad." + k, g[k], True)
    np.savez_compressed(os.path.join(GOLD, "inpaint_b1.npz"), **st)


# ---- 4. CRN_smaller ---------------------------------------------------------------------------------
def g_crn():
    for sp, B, full in ((64, 2, True), (256, 1, False)):
        st = {}
        m = synth.load_synth(RCRN(3, fg=True), 41)
        x = T(synth.uniform(41, "label%d" % sp, (B, 3, sp, sp))).requires_grad_(True)
        rgb, mask = m(x, sp)
        o_rgb, o_mask = O.crn_smaller_forward(sd_of(m), x.detach(), sp, True)
        report["crn%d" % sp] = close(o_rgb, rgb, 5e-4, "CRN_smaller rgb sp=%d" % sp)
        close(o_mask, mask, 1e-4, "CRN_smaller mask sp=%d" % sp)
        put(st, "rgb", rgb, full); put(st, "mask", mask, full)
        if full:
            proj = T(synth.uniform(41, "proj", tuple(rgb.shape)))
            g = grads_of(m, (rgb * proj).sum() + mask.sum())
            put(st, "dlabel", x.grad, True)
            for k in ["conv1_encoder.conv_block.0.weight", "conv6_decoder.conv_block.1.gamma", "decoder.conv_block.4.beta",
                      "out_conv.weight", "fg_conv.bias", "conv3_decoder.conv_block.3.bias"]:
                put(st, "grad." + k, g[k], k != "conv1_encoder.conv_block.0.weight" or True)
        np.savez_compressed(os.path.join(GOLD, "crn_sp%d.npz" % sp), **st)


# ---- 5. Propagation3DFlowNet, train-mode BN (F9) -----------------------------------------------------
def pro_inputs(seed, B, S):
    return {"fake_tgt": T(synth.uniform(seed, "fake", (B, 3, S, S))).requires_grad_(True),
            "tsf_image": T(synth.uniform(seed, "tsf", (B, 3, S, S))),
            "tgt_smpl_mask": T((synth.uniform(seed, "mask", (B, 3, S, S)) > 0).astype(np.float32)),
            "tgt_IUV": T(synth.uniform(seed, "iuv", (B, 3, S, S))), "use_mask": True, "use_IUV": True}


def g_propagation():
    st = {}
    m = synth.load_synth(RPro(9, 32, 2, 3, use_deconv=False), 51)
    m.train()
    sd0 = sd_of(m)
    x = pro_inputs(51, 2, 64)
    out = m(x)
    proj = T(synth.uniform(51, "proj", tuple(out["pred_target"].shape)))
    g = grads_of(m, (out["pred_target"] * proj).sum())
    oo = O.propagation_forward(sd0, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in x.items()}, True)
    report["propagation"] = close(oo["pred_target"], out["pred_target"], 2e-5, "Propagation3DFlowNet pred")
    close(oo["weight"], out["weight"], 2e-5, "Propagation3DFlowNet weight")
    sd1 = sd_of(m)
    close(sd0["composite_unet.model_down_img.2.running_var"], sd1["composite_unet.model_down_img.2.running_var"], 1e-6,
          "BN running_var after step")
    put(st, "pred", out["pred_target"], True); put(st, "weight", out["weight"], True)
    put(st, "dfake", x["fake_tgt"].grad, True)
    for k in ["composite_unet.model_down_img.2.running_mean", "composite_unet.model_down_img.2.running_var",
              "composite_unet.model_up_img.6.running_var", "composite_unet.model_res_img.0.conv_block.6.running_mean"]:
        put(st, "after." + k, sd1[k], True)
    for k in ["composite_unet.model_down_img.1.weight", "composite_unet.model_down_img.10.conv_block.2.weight",
              "composite_unet.model_final_w.1.weight", "composite_unet.model_up_img.5.bias"]:
        put(st, "grad." + k, g[k], True)
    np.savez_compressed(os.path.join(GOLD, "propagation_64.npz"), **st)


# ---- 6. discriminators ------------------------------------------------------------------------------
def g_disc():
    st = {}
    for name, cls, size, convs in (("D", RN.ImageDiscriminator, 256, O.IMAGE_D_CONVS), ("FD", RN.FaceDiscriminator, 64, O.FACE_D_CONVS)):
        m = synth.load_synth(cls(32, 6), 61)
        m.train()
        sd0 = sd_of(m)
        x = T(synth.uniform(61, name + "x", (2, 6, size, size))).requires_grad_(True)
        p = m(x)
        loss = torch.nn.functional.binary_cross_entropy(p, torch.ones_like(p))
        g = grads_of(m, loss)
        po = O.discriminator_forward(sd0, x.detach(), True, convs)
        report["disc_" + name] = close(po, p, 1e-5, name + " prob")
        put(st, name + ".p", p, True); put(st, name + ".loss", loss.reshape(1), True); put(st, name + ".dx", x.grad)
        put(st, name + ".grad.main.0.weight", g["main.0.weight"], True)
        put(st, name + ".grad.classifier.2.weight", g["classifier.2.weight"], True)
        put(st, name + ".grad.main.3.weight", g["main.3.weight"], True)
        put(st, name + ".after.main.3.running_var", sd_of(m)["main.3.running_var"], True)
    np.savez_compressed(os.path.join(GOLD, "discriminators.npz"), **st)


# ---- 7. texture warp (train/4...py:43-76, float-patched torch.full per F6 is not involved here) ------
def g_texwarp():
    sys.path.insert(0, "/root/reference/train")
    st = {}
    iuv = synth.iuv255(71, "iuv", 1, 256)[0]
    tex = [T(synth.uniform(71, "tex%d" % p, (3, 200, 200))) for p in range(24)]
    # reference function body uses the torch default align_corners (False under torch 2.x, F7)
    ref = RN.texture_warp_pytorch(tex, torch.from_numpy(iuv))
    mine = O.texture_warp(tex, iuv, align_corners=False)
    report["texture_warp"] = close(mine, ref, 1e-6, "texture_warp_pytorch")
    put(st, "out_ac0", ref)
    put(st, "out_ac1", O.texture_warp(tex, iuv, align_corners=True))
    np.savez_compressed(os.path.join(GOLD, "texture_warp.npz"), **st)


# ---- 8. flow: projection + rasteriser (C oracle) + cal_bc_transform (reference method) + grid_sample --
def g_flow():
    import neural_renderer_stub  # noqa: F401  (installed below)
    from src.nmr import SMPLRenderer, orthographic_proj_withz_idrot
    st = {}
    B = 2
    verts_s, verts_t = T(synth.posed_vertices(81, "src", B)), T(synth.posed_vertices(81, "tgt", B))
    cam = torch.zeros(B, 3); cam[:, 0] = 0.9
    _, fidx = synth.body_mesh()
    # reference projection chain (src/nmr.py:269-276) with the real look_at / vertices_to_faces
    import neural_renderer as nr
    def ref_faces(v):
        pv = orthographic_proj_withz_idrot(v, cam)
        pv[:, :, 1] *= -1
        pv = nr.look_at(pv, [0, 0, O.EYE_Z])
        return nr.vertices_to_faces(pv, T(fidx)[None].repeat(B, 1, 1))
    fs_ref, ft_ref = ref_faces(verts_s), ref_faces(verts_t)
    fs, ft = O.project_faces(verts_s, cam, fidx), O.project_faces(verts_t, cam, fidx)
    report["project_faces"] = close(fs, fs_ref, 1e-6, "projection+look_at+v2f")
    fim, wim = raster_oracle.rasterize_fim_wim(ft.numpy(), 256)
    cov = float((fim >= 0).mean())
    print("  body mesh coverage %.3f" % cov)
    f2 = fs_ref[:, :, :, 0:2].clone(); f2[:, :, :, 1] *= -1
    fake_self = types.SimpleNamespace(image_size=256)
    T_ref = SMPLRenderer.cal_bc_transform(fake_self, f2, T(fim), T(wim))
    src_img = T(synth.uniform(81, "img", (B, 3, 256, 256)))
    warped_ref = torch.nn.functional.grid_sample(src_img, T_ref, padding_mode='border')
    warped, T_mine = O.flow_warp(src_img, fs, T(fim), T(wim))
    report["bc_transform"] = close(T_mine, T_ref, 1e-6, "cal_bc_transform")
    close(warped, warped_ref, 1e-5, "flow warp")
    st["coverage"] = np.float64(cov)
    put(st, "faces_tgt", ft_ref); put(st, "T", T_ref); put(st, "warped", warped_ref)
    st["fim.sum"] = np.int64(fim.astype(np.int64).sum()); st["fim.cov"] = np.int64((fim >= 0).sum())
    idx = np.linspace(0, fim.size - 1, 4096).astype(np.int64)
    st["fim.idx"] = idx; st["fim.samples"] = fim.reshape(-1)[idx]
    put(st, "wim", T(wim))
    np.savez_compressed(os.path.join(GOLD, "flow_b2.npz"), **st)


def install_nr_stub():
    import importlib.util
    nr = types.ModuleType("neural_renderer")
    base = "/root/reference/third_party/neural_renderer/neural_renderer/"
    for name in ("look_at", "vertices_to_faces"):
        spec = importlib.util.spec_from_file_location("nr_" + name, base + name + ".py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        setattr(nr, name, getattr(mod, name))
    sys.modules["neural_renderer"] = nr
    sys.modules["neural_renderer_stub"] = nr
    np.float = float


# ---- 9. VGG_l1_loss on a synthetic VGG19 (in-place ReLU, F8) ------------------------------------------
def g_vgg():
    import torch.nn as nn
    st = {}
    layers, cin = [], 3
    for v in O.VGG_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    feats = nn.Sequential(*layers)
    RN.vgg19 = lambda pretrained=True: types.SimpleNamespace(features=feats)
    m = RN.VGG_l1_loss()
    synth.load_synth(m, 91)
    x = T(synth.uniform(91, "x", (1, 3, 64, 64))).requires_grad_(True)
    y = T(synth.uniform(91, "y", (1, 3, 64, 64)))
    loss = m(x, y)
    loss.backward()
    lo = O.vgg_l1_loss(sd_of(m), x.detach(), y)
    report["vgg_l1"] = close(lo.reshape(1), loss.reshape(1), 1e-3 * max(1.0, abs(loss.item())) * 1e-2, "VGG_l1_loss")
    put(st, "loss", loss.reshape(1), True); put(st, "dx", x.grad, True)
    np.savez_compressed(os.path.join(GOLD, "vgg_l1_64.npz"), **st)


# ---- 10. stage-1 Accumulate_LSTM loss (BASELINE config 1) ----------------------------------------------
def g_stage1():
    st = {}
    m = synth.load_synth(RN.Accumulate_LSTM(), 101)
    x_in = accu_inputs(101, 1, 2)
    src_mask = T(synth.rect_masks(101, "sm", (1, 2, 3, 800, 1200)).astype(np.uint8))
    tgt_mask = T(synth.rect_masks(101, "tm", (1, 3, 3, 800, 1200)).astype(np.uint8))
    tgt_tex = T(synth.uniform(101, "tt", (1, 3, 3, 800, 1200)))
    atlas, loss = m(x_in, src_mask, tgt_mask, tgt_tex)
    g = grads_of(m, loss)
    a2, l2 = O.accumulate_lstm_loss(sd_of(m), x_in, src_mask, tgt_mask, tgt_tex)
    report["stage1_loss"] = close(l2.reshape(1), loss.reshape(1), 1e-6, "Accumulate_LSTM loss")
    close(a2, atlas, 2e-5, "Accumulate_LSTM atlas")
    put(st, "loss", loss.reshape(1), True); put(st, "atlas", atlas)
    put(st, "grad.Upsampler_list.3.conv.weight", g["Upsampler_list.3.conv.weight"], True)
    np.savez_compressed(os.path.join(GOLD, "stage1_b1_t2.npz"), **st)


# ---- 10b. BASELINE config 1 in full: stage-1 step at T=4 with backward + Adam(1e-4) (train/1...py:140-176) -------
def g_stage1_t4():
    from oracle.stage_oracle import OracleStage1
    st = {}
    m = synth.load_synth(RN.Accumulate_LSTM(), 111)
    sd0 = sd_of(m)
    b = {k: T(v) for k, v in synth.stage1_batch(611, 1).items()}
    used = [0, 1, 2, 3]
    x_in = [[b["src_texture_im"][:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200] for t in used] for i in range(4) for j in range(6)]
    src_mask = b["src_mask_im"].byte().unsqueeze(2).repeat(1, 1, 3, 1, 1)
    tgt_mask = b["tgt_mask_im"].byte().unsqueeze(2).repeat(1, 1, 3, 1, 1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
    opt.zero_grad()
    atlas, loss = m(x_in, src_mask, tgt_mask, b["tgt_texture_im"])
    loss.sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    opt.step()
    after = sd_of(m)
    orc = OracleStage1(sd0)
    ro = orc.train_step(b, used)
    report["stage1_t4_loss"] = close(ro["total_loss"].reshape(1), loss.reshape(1), 1e-6, "stage-1 T=4 loss")
    num = den = 0.0
    for k, g in grads.items():
        num += float(((orc.sd[k].grad - g).double() ** 2).sum()); den += float((g.double() ** 2).sum())
    report["stage1_t4_grad_rel"] = (num / den) ** 0.5
    print("  oracle vs reference stage-1 T=4 gradients rel-L2 = %.3e" % report["stage1_t4_grad_rel"])
    assert report["stage1_t4_grad_rel"] <= 1e-5
    report["stage1_t4_adam_max"] = max(float((orc.sd[k].detach() - v).abs().max()) for k, v in after.items())
    assert report["stage1_t4_adam_max"] <= 2.01e-4        # lr * sign(g) apart at worst (a ~zero gradient flipping sign)
    put(st, "loss", loss.reshape(1), True); put(st, "atlas", atlas)
    keys = ["Downsampler_list.0.enc1.enconv.0.weight", "Downsampler_list.5.convLSTM1.cell_list.0.conv.weight",
            "Downsampler_list.11.convLSTM5.cell_list.0.conv.bias", "Downsampler_list.17.enc8.enconv.0.weight",
            "Upsampler_list.3.dec2.myconv.0.weight", "Upsampler_list.23.conv.weight"]
    for k in keys:
        put(st, "grad." + k, grads[k], True)
        put(st, "delta." + k, after[k] - sd0[k], True)
    # whole-model digests: sum of squares of every gradient, per parameter family
    fam = {}
    for k, g in grads.items():
        f = ".".join(k.split(".")[2:])
        fam[f] = fam.get(f, 0.0) + float((g.double() ** 2).sum())
    st["gradsq.families"] = np.array(sorted(fam), dtype="U64")
    st["gradsq.values"] = np.array([fam[f] for f in sorted(fam)], np.float64)
    np.savez_compressed(os.path.join(GOLD, "stage1_b1_t4_step.npz"), **st)


# ---- 10c. checkpoint files go both ways (train/3...py:481-494, train/4...py:121-140,518-533) --------------------
def g_checkpoints():
    import hashlib
    import tempfile
    from jafpro_amd import crn_model, flow_net, networks, stages
    pairs = {"accu": (RN.Accumulate_LSTM_no_loss, networks.Accumulate_LSTM_no_loss), "inpaint": (RN.UNet_inpainter, networks.UNet_inpainter),
             "bg": (lambda: RCRN(3), lambda: crn_model.CRN_smaller(3)), "refine": (lambda: RCRN(3, fg=True), lambda: crn_model.CRN_smaller(3, fg=True)),
             "D": (lambda: RN.ImageDiscriminator(32, 6), lambda: networks.ImageDiscriminator(32, 6)),
             "face": (lambda: RN.FaceDiscriminator(32, 6), lambda: networks.FaceDiscriminator(32, 6)),
             "flow": (lambda: RPro(9, 32, 2, 3, use_deconv=False), lambda: flow_net.Propagation3DFlowNet(9, 32, 2, 3, use_deconv=False))}
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for i, (name, (mk_ref, mk_mir)) in enumerate(pairs.items()):
            ref, mir = synth.load_synth(mk_ref(), 700 + i), mk_mir()
            # reference -> file -> mirror (what loading the released Accu_/inpaint_/bg_/refine_/pro_iter_*.pth does)
            rp = os.path.join(td, "ref_%s.pth" % name)
            torch.save(ref.state_dict(), rp)
            res = stages.load_checkpoint(mir, rp, strict=True)
            ok_in = not res.missing_keys and not res.unexpected_keys and all(
                torch.equal(a, b) for a, b in zip(ref.state_dict().values(), mir.state_dict().values()))
            # mirror -> file -> reference
            mp = stages.save_checkpoints(td, 36000, {name: mir})[name]
            assert os.path.basename(mp) == "%s_iter_36000.pth" % stages.CKPT_PREFIX[name]
            ref2 = mk_ref()
            res2 = ref2.load_state_dict(torch.load(mp), strict=True)
            ok_out = all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), ref2.state_dict().values()))
            keys_same = list(torch.load(mp).keys()) == list(ref.state_dict().keys())
            h = hashlib.sha256()
            for k, v in ref.state_dict().items():
                h.update(k.encode()); h.update(v.detach().cpu().contiguous().numpy().tobytes())
            out[name] = {"file": os.path.basename(mp), "reference_to_mirror": bool(ok_in), "mirror_to_reference": bool(ok_out),
                         "key_order_equal": bool(keys_same), "entries": len(ref.state_dict()), "seed": 700 + i, "sha256": h.hexdigest()}
            print("  checkpoint %-8s in %s out %s order %s" % (name, ok_in, ok_out, keys_same))
            assert ok_in and ok_out and keys_same
    json.dump(out, open(os.path.join(GOLD, "checkpoint_pin.json"), "w"), indent=1, sort_keys=True)
    report["checkpoint_files_roundtrip"] = len(out)


# ---- 12. FlowNetSD of the evaluation script (test/video_evaluation.py:66,197-206) ----------------------
def g_flownet():
    from src.flownet2_pytorch.networks.FlowNetSD import FlowNetSD as RFlowNetSD
    ref = RFlowNetSD(args=[], batchNorm=False)            # left in train mode, as the script does: forward returns 5 flows
    synth.load_synth(ref, 811)
    st = {}
    # two frame pairs of a synthetic video in (-1, 1), through flownet_preprocess (:34-36)
    frames = synth.uniform(812, "flow_frames", (3, 3, 128, 192))
    smooth = torch.nn.functional.avg_pool2d(T(frames), 5, 1, 2)                  # some spatial structure, still seeded
    pairs = torch.cat([smooth[:-1], smooth[1:]], 1) / 2.0 + 0.5
    with torch.no_grad():
        flows = ref(pairs)
        mine = O.flownet_sd_forward(sd_of(ref), pairs, True)
    assert len(flows) == 5
    for i, (a, b) in enumerate(zip(flows, mine)):
        report["flownet_sd_flow%d" % (i + 2)] = close(b, a, 0.0, "FlowNetSD flow%d" % (i + 2))
    st["pairs"] = pairs.numpy()
    for i, a in enumerate(flows):
        put(st, "flow%d" % (i + 2), a, full=True)
    ref.eval()
    with torch.no_grad():
        assert len(ref(pairs)) == 1 and torch.equal(ref(pairs)[0], flows[0])
    st["flow_l1"] = np.float64(torch.nn.functional.l1_loss(flows[0][:1], flows[0][1:]).item())   # the script's per-frame term
    np.savez_compressed(os.path.join(GOLD, "flownet_sd.npz"), **st)
    schema_path = os.path.join(GOLD, "state_dict_schema.json")
    schema = json.load(open(schema_path))
    schema["FlowNetSD"] = [[k, list(v.shape)] for k, v in ref.state_dict().items()]
    json.dump(schema, open(schema_path, "w"))


# ---- 13. TransferTexture of the data pipeline (src/utils.py:369-394) ---------------------------------
def g_data():
    """src/utils.py cannot be imported (tensorflow, cv2, moviepy at module level), but TransferTexture itself is pure
    NumPy: the function definition is taken out of the module's syntax tree and executed here, in memory, with `np` as its
    only global -- the reference's own code producing the expected outputs of oracle/data_oracle.transfer_texture."""
    import ast
    from oracle import data_oracle
    path = "/root/reference/src/utils.py"
    tree = ast.parse(open(path).read(), filename=path)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "TransferTexture"]
    assert len(fn) == 1
    ns = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    ref_fn = ns["TransferTexture"]
    raw = synth.stage4_raw(821, 3)
    st, worst = {}, 0
    for i in range(3):
        tex, iuv, im = raw["src_texture_u8"][i, 0], raw["tgt_IUV_u8"][i], raw["tgt_img_u8"][i]
        for tag, bg in (("plain", None), ("over_image", im)):
            ref = ref_fn(tex.copy(), iuv.copy(), None if bg is None else bg.copy())
            mine = data_oracle.transfer_texture(tex, iuv, bg)
            assert ref.dtype == np.uint8 and ref.shape == (256, 256, 3)
            worst = max(worst, int(np.abs(ref.astype(np.int64) - mine.astype(np.int64)).max()))
            st["%s.%d" % (tag, i)] = ref
    ones = ref_fn(np.ones((800, 1200, 3), np.uint8), raw["src_IUV0_u8"][0].copy())          # the mask use of src/data.py:690-695
    assert np.array_equal(ones, data_oracle.transfer_texture(np.ones((800, 1200, 3), np.uint8), raw["src_IUV0_u8"][0]))
    st["ones_mask.0"] = ones
    st["seed"] = np.int64(821)
    print("  oracle vs reference TransferTexture          max|diff| = %d" % worst)
    assert worst == 0
    report["transfer_texture"] = float(worst)
    np.savez_compressed(os.path.join(GOLD, "transfer_texture.npz"), **st)


# ---- 14. UV-map asset builders of SMPLRenderer's static-UV branch (src/mesh.py:28-77,156-194,368-423,530-568) ----
def g_mesh():
    """src/mesh.py imports only numpy / torch / json: the reference's own create_uvsampler / create_mapping run here on
    synthetic assets in the reference's file formats (jafpro_amd/synth.uv_assets: the real mapper.txt is not redistributable)
    and pin jafpro_amd/mesh.py bit for bit."""
    import tempfile
    import src.mesh as RM
    from jafpro_amd import mesh as M
    st, checked = {}, 0
    with tempfile.TemporaryDirectory() as td:
        a = synth.uv_assets(td, seed=7)
        for T_ in (2, 3, 6):
            ref = RM.create_uvsampler(a["obj"], tex_size=T_)
            mine = M.create_uvsampler(a["obj"], tex_size=T_)
            assert ref.dtype == mine.dtype and np.array_equal(ref, mine), ("uvsampler", T_)
            st["uvsampler.%d" % T_] = ref
            checked += 1
        ro, mo = RM.load_obj(a["obj"]), M.load_obj(a["obj"])
        for k in ro:
            assert ro[k].dtype == mo[k].dtype and np.array_equal(ro[k], mo[k]), k
        for fb in (False, True):
            assert np.array_equal(RM.get_f2vts(a["obj"], fill_back=fb), M.get_f2vts(a["obj"], fill_back=fb))
            for name in ("uv", "seg", "uv_seg", "par", "front", "head", "back", "binary"):
                if name == "par" and fb:
                    continue        # (the reference's par mapping ignores fill_back and then fails its own face count)
                kw = dict(part_info=a["part_info"], front_info=a["front_info"], head_info=a["head_info"], contain_bg=True, fill_back=fb)
                ref = RM.create_mapping(name, a["obj"], **kw)
                mine = M.create_mapping(name, a["obj"], **kw)
                assert ref.dtype == mine.dtype and ref.shape == mine.shape and np.array_equal(ref, mine), (name, fb)
                st["map.%s.%d" % (name, int(fb))] = ref
                checked += 1
        ref = RM.create_mapping("ids", a["obj"], contain_bg=False)           # (with the background row the reference raises)
        assert np.array_equal(ref, M.create_mapping("ids", a["obj"], contain_bg=False))
        st["map.ids.nobg"] = ref
        st["nf"] = np.int64(a["nf"])
        st["seed"] = np.int64(7)
    print("  jafpro_amd.mesh vs reference src/mesh.py       %d tables bit-identical" % checked)
    report["mesh_assets_tables"] = float(checked)
    np.savez_compressed(os.path.join(GOLD, "mesh_assets.npz"), **st)


# ---- 11. state_dict schema of the boundary modules -------------------------------
def g_schema():
    mods = {"Accumulate_LSTM_no_loss": RN.Accumulate_LSTM_no_loss(), "UNet_inpainter": RN.UNet_inpainter(),
            "CRN_smaller_fg": RCRN(3, fg=True), "CRN_smaller": RCRN(3),
            "Propagation3DFlowNet": RPro(9, 32, 2, 3, use_deconv=False),
            "ImageDiscriminator": RN.ImageDiscriminator(32, 6), "FaceDiscriminator": RN.FaceDiscriminator(32, 6),
            "ConvLSTM": RConvLSTM((7, 5), 4, [4], [(3, 3)], 1, batch_first=True, bias=True)}
    schema = {name: [[k, list(v.shape)] for k, v in m.state_dict().items()] for name, m in mods.items()}
    json.dump(schema, open(os.path.join(GOLD, "state_dict_schema.json"), "w"))
    report["schema_modules"] = len(schema)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    install_nr_stub()
    which = sys.argv[1:] or ["convlstm", "accumulate", "inpaint", "crn", "propagation", "disc", "texwarp", "flow", "vgg", "stage1"]
    for w in which:
        print("==", w)
        globals()["g_" + w]()
    prev = {}
    rp = os.path.join(GOLD, "oracle_pin_report.json")
    if os.path.exists(rp):
        prev = json.load(open(rp))
    prev.update(report)
    json.dump(prev, open(rp, "w"), indent=1, sort_keys=True)
    print(json.dumps(report, indent=1))
