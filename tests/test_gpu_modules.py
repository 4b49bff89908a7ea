"""GPU parity of the nn.Module mirrors against the committed golden vectors, which were produced
by the REFERENCE modules (oracle/make_golden.py) on the same portable synthetic weights/inputs.
Forward bar: <= 1e-3 L-inf in fp32 (BASELINE.json north_star); gradients: 1e-3 relative."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
FWD_TOL = 1e-3


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def check(st, name, t, tol, rel=False):
    t = t.detach().float().cpu()
    if name in st:
        ref = torch.from_numpy(st[name])
        assert tuple(ref.shape) == tuple(t.shape), (name, ref.shape, t.shape)
        err = (t - ref).abs().max().item()
        scale = max(1.0, ref.abs().max().item()) if rel else 1.0
        assert err <= tol * scale, "%s: max|diff| %.3e > %.1e*%.2f" % (name, err, tol, scale)
        return err
    assert tuple(st[name + ".shape"]) == tuple(t.shape), name
    flat = t.reshape(-1)
    ref = torch.from_numpy(st[name + ".samples"])
    err = (flat[torch.from_numpy(st[name + ".idx"])] - ref).abs().max().item()
    scale = max(1.0, ref.abs().max().item()) if rel else 1.0
    assert err <= tol * scale, "%s: sample max|diff| %.3e" % (name, err)
    s = flat.double().sum().item()
    sa = flat.double().abs().sum().item()
    assert abs(sa - st[name + ".sumabs"]) <= max(tol * flat.numel() * 0.05, 1e-3 * abs(st[name + ".sumabs"])), name
    assert abs(s - st[name + ".sum"]) <= max(tol * flat.numel() * 0.05, 1e-3 * abs(st[name + ".sumabs"])), name
    return err


def grouped_grad(module, key):
    """gradient of a reference-keyed per-part parameter from the grouped parameter's .grad"""
    for pname, template in module._key_map.items():
        for p in range(24):
            if template.format(p=p) == key:
                g = getattr(module, pname).grad
                per = g.shape[0] // 24
                return g[p * per:(p + 1) * per]
    raise KeyError(key)


def named_grad(module, key):
    return dict(module.named_parameters())[key].grad


def test_convlstm_toy(golden_dir):
    from jafpro_amd import synth
    from jafpro_amd.convLSTM import ConvLSTM
    st = load(golden_dir, "convlstm_toy.npz")
    m = synth.load_synth(ConvLSTM((7, 5), 4, [4], [(3, 3)], 1, batch_first=True, bias=True), 11).cuda()
    x = T(synth.uniform(11, "x", (2, 3, 4, 7, 5))).requires_grad_(True)
    out, last = m(x)
    check(st, "out", out, 1e-5); check(st, "h_T", last[0][0], 1e-5); check(st, "c_T", last[0][1], 1e-5)
    proj = T(synth.uniform(11, "proj", tuple(last[0][0].shape)))
    ((last[0][0] * proj).sum() + 0.5 * (out * out).sum()).backward()
    check(st, "dx", x.grad, 1e-4)
    check(st, "dw", m.cell_list[0].conv.weight.grad, 1e-3, rel=True)
    check(st, "db", m.cell_list[0].conv.bias.grad, 1e-3, rel=True)


def test_accumulate_golden(golden_dir):
    from jafpro_amd import synth
    from jafpro_amd.networks import Accumulate_LSTM_no_loss
    st = load(golden_dir, "accumulate_b1_t2.npz")
    m = synth.load_synth(Accumulate_LSTM_no_loss(), 21).cuda()
    atlas = synth.uniform(21, "src_texture_im", (1, 2, 3, 800, 1200))
    x_in = [[T(atlas[:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200]) for t in range(2)]
            for i in range(4) for j in range(6)]
    outs = m(x_in)
    assert len(outs) == 24 and tuple(outs[0].shape) == (1, 3, 200, 200)
    out = torch.cat(outs, 1)
    check(st, "out", out, FWD_TOL)
    proj = T(synth.uniform(21, "proj", tuple(out.shape)))
    (out * proj).sum().backward()
    for k in [k[5:] for k in st if k.startswith("grad.")]:
        check(st, "grad." + k, grouped_grad(m, k), 2e-3, rel=True)


def test_inpaint_golden(golden_dir):
    from jafpro_amd import synth
    from jafpro_amd.networks import UNet_inpainter
    st = load(golden_dir, "inpaint_b1.npz")
    m = synth.load_synth(UNet_inpainter(), 31).cuda()
    tex = [T(synth.uniform(31, "tex%d" % p, (1, 3, 200, 200))).requires_grad_(True) for p in range(24)]
    out = torch.cat(m(tex), 1)
    check(st, "out", out, FWD_TOL)
    proj = T(synth.uniform(31, "proj", tuple(out.shape)))
    (out * proj).sum().backward()
    check(st, "dtex3", tex[3].grad, 2e-3, rel=True)
    for k in [k[5:] for k in st if k.startswith("grad.")]:
        check(st, "grad." + k, grouped_grad(m, k), 2e-3, rel=True)


@pytest.mark.parametrize("sp,B", [(64, 2), (256, 1)])
def test_crn_golden(golden_dir, sp, B):
    from jafpro_amd import synth
    from jafpro_amd.crn_model import CRN_smaller
    st = load(golden_dir, "crn_sp%d.npz" % sp)
    m = synth.load_synth(CRN_smaller(3, fg=True), 41).cuda()
    x = T(synth.uniform(41, "label%d" % sp, (B, 3, sp, sp))).requires_grad_(True)
    rgb, mask = m(x, sp)
    check(st, "rgb", rgb, FWD_TOL); check(st, "mask", mask, FWD_TOL)
    if sp == 64:
        proj = T(synth.uniform(41, "proj", tuple(rgb.shape)))
        ((rgb * proj).sum() + mask.sum()).backward()
        check(st, "dlabel", x.grad, 2e-3, rel=True)
        for k in [k[5:] for k in st if k.startswith("grad.")]:
            check(st, "grad." + k, named_grad(m, k), 2e-3, rel=True)


def test_propagation_golden(golden_dir):
    from jafpro_amd import synth
    from jafpro_amd.flow_net import Propagation3DFlowNet
    st = load(golden_dir, "propagation_64.npz")
    m = synth.load_synth(Propagation3DFlowNet(9, 32, 2, 3, use_deconv=False), 51).cuda()
    m.train()
    B, S = 2, 64
    x = {"fake_tgt": T(synth.uniform(51, "fake", (B, 3, S, S))).requires_grad_(True),
         "tsf_image": T(synth.uniform(51, "tsf", (B, 3, S, S))),
         "tgt_smpl_mask": T((synth.uniform(51, "mask", (B, 3, S, S)) > 0).astype(np.float32)),
         "tgt_IUV": T(synth.uniform(51, "iuv", (B, 3, S, S))), "use_mask": True, "use_IUV": True}
    out = m(x)
    check(st, "pred", out["pred_target"], FWD_TOL); check(st, "weight", out["weight"], FWD_TOL)
    sd = m.state_dict()
    for k in [k[6:] for k in st if k.startswith("after.")]:
        check(st, "after." + k, sd[k], 1e-5)                      # train-mode BN side effect (F9)
    assert int(sd["composite_unet.model_down_img.2.num_batches_tracked"]) == 1
    proj = T(synth.uniform(51, "proj", tuple(out["pred_target"].shape)))
    (out["pred_target"] * proj).sum().backward()
    check(st, "dfake", x["fake_tgt"].grad, 2e-3, rel=True)
    for k in [k[5:] for k in st if k.startswith("grad.")]:
        check(st, "grad." + k, named_grad(m, k), 2e-3, rel=True)


def test_discriminators_golden(golden_dir):
    from jafpro_amd import ops, synth
    from jafpro_amd.networks import FaceDiscriminator, ImageDiscriminator
    st = load(golden_dir, "discriminators.npz")
    for name, cls, size in (("D", ImageDiscriminator, 256), ("FD", FaceDiscriminator, 64)):
        m = synth.load_synth(cls(32, 6), 61).cuda()
        m.train()
        x = T(synth.uniform(61, name + "x", (2, 6, size, size))).requires_grad_(True)
        p = m(x)
        check(st, name + ".p", p, 1e-4)
        loss = ops.bce_loss(p, 1.0)
        check(st, name + ".loss", loss, 1e-4)
        loss.sum().backward()
        check(st, name + ".dx", x.grad, 2e-3, rel=True)
        for k in ("main.0.weight", "classifier.2.weight", "main.3.weight"):
            check(st, name + ".grad." + k, named_grad(m, k), 2e-3, rel=True)
        check(st, name + ".after.main.3.running_var", m.state_dict()["main.3.running_var"], 1e-5)


def test_texture_warp_golden(golden_dir):
    from jafpro_amd import synth
    from jafpro_amd.networks import texture_warp_pytorch
    st = load(golden_dir, "texture_warp.npz")
    iuv = synth.iuv255(71, "iuv", 1, 256)[0]
    tex = [T(synth.uniform(71, "tex%d" % p, (3, 200, 200))) for p in range(24)]
    check(st, "out_ac0", texture_warp_pytorch(tex, iuv, "cuda"), 1e-5)
    check(st, "out_ac1", texture_warp_pytorch(tex, iuv, "cuda", align_corners=True), 1e-5)


def test_flow_golden(golden_dir):
    from jafpro_amd import synth
    from jafpro_amd.cal_flow import float_estimate
    st = load(golden_dir, "flow_b2.npz")
    B = 2
    _, fidx = synth.body_mesh()
    fe = float_estimate(faces=fidx).cuda()
    vs, vt = T(synth.posed_vertices(81, "src", B)), T(synth.posed_vertices(81, "tgt", B))
    cam = torch.zeros(B, 3).cuda(); cam[:, 0] = 0.9
    ft, fim, wim = fe.render.render_fim_wim(cam, vt)
    check(st, "faces_tgt", ft, 1e-6)
    f = fim.cpu().numpy()
    assert int((f >= 0).sum()) == int(st["fim.cov"]) and int(f.astype(np.int64).sum()) == int(st["fim.sum"])
    assert (f.reshape(-1)[st["fim.idx"]] == st["fim.samples"]).all()
    check(st, "wim", wim, 1e-7)
    src_img = T(synth.uniform(81, "img", (B, 3, 256, 256)))
    flow = fe.cal_flow(cam, None, vs, None, cam, None, vt, None)
    check(st, "T", flow, 1e-6)
    check(st, "warped", fe(src_img, [cam, None, vs, None], [cam, None, vt, None]), 1e-5)


def test_vgg_l1_golden(golden_dir):
    from jafpro_amd import synth
    from jafpro_amd.networks import VGG_l1_loss
    st = load(golden_dir, "vgg_l1_64.npz")
    m = VGG_l1_loss()
    # the golden VGG was filled through the reference's state_dict, which also holds conv5_3/5_4
    sd = m.state_dict()
    vals = synth.synth_state_dict({k: tuple(v.shape) for k, v in sd.items()}, 91)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    m = m.cuda()
    x = T(synth.uniform(91, "x", (1, 3, 64, 64))).requires_grad_(True)
    y = T(synth.uniform(91, "y", (1, 3, 64, 64)))
    loss = m(x, y)
    ref = float(st["loss"][0])
    assert abs(loss.item() - ref) <= 1e-3 * max(1.0, abs(ref)), (loss.item(), ref)
    loss.backward()
    # d|a-b| = sign(a-b) is discontinuous: a feature difference of 1e-6 flips a sign and moves single
    # pixels of dx by O(1e-2); gradients of the L1 terms are therefore compared in relative L2.
    ref_dx = torch.from_numpy(st["dx"])
    rel = ((x.grad.cpu() - ref_dx).norm() / ref_dx.norm()).item()
    print("vgg dx rel-L2 %.3e" % rel)
    assert rel <= 2e-2, rel


def test_stage1_golden(golden_dir):
    """BASELINE config 1: Accumulate_LSTM with atlas paste + masked L1 (src/networks.py:1607-1639)."""
    from jafpro_amd import synth
    from jafpro_amd.networks import Accumulate_LSTM
    st = load(golden_dir, "stage1_b1_t2.npz")
    m = synth.load_synth(Accumulate_LSTM(), 101).cuda()
    atlas = synth.uniform(101, "src_texture_im", (1, 2, 3, 800, 1200))
    x_in = [[T(atlas[:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200]) for t in range(2)]
            for i in range(4) for j in range(6)]
    src_mask = T(synth.rect_masks(101, "sm", (1, 2, 3, 800, 1200)).astype(np.uint8))
    tgt_mask = T(synth.rect_masks(101, "tm", (1, 3, 3, 800, 1200)).astype(np.uint8))
    tgt_tex = T(synth.uniform(101, "tt", (1, 3, 3, 800, 1200)))
    atlas_out, loss = m(x_in, src_mask, tgt_mask, tgt_tex)
    check(st, "atlas", atlas_out, FWD_TOL)
    assert abs(loss.item() - float(st["loss"][0])) <= 1e-4
    loss.backward()
    check(st, "grad.Upsampler_list.3.conv.weight", grouped_grad(m, "Upsampler_list.3.conv.weight"), 2e-3, rel=True)


def test_state_dict_roundtrip_on_gpu():
    from jafpro_amd import synth
    from jafpro_amd.networks import UNet_inpainter
    a = synth.load_synth(UNet_inpainter(), 5).cuda()
    b = UNet_inpainter().cuda()
    b.load_state_dict(a.state_dict())
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


def test_ops_reject_cpu_tensors():
    from jafpro_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, pad=1)
    with pytest.raises(RuntimeError):
        ops.avg_pool(torch.zeros(1, 1, 4, 4), 2, 2, 0)


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 6e-2)])
def test_convlstm_cell_single_step_and_explicit_state(mode, tol):
    """ConvLSTMCell.forward(input, (h, c)) -> (h_next, c_next) (src/convLSTM.py:41-56) and ConvLSTM.forward with a passed
    hidden_state (:119-128), against the oracle cell on CPU: values, and gradients w.r.t. the input, the initial
    state (h0 AND c0), the weights and the bias, with a loss that uses both h_T and c_T."""
    from jafpro_amd import ops, synth
    from jafpro_amd.convLSTM import ConvLSTM
    from oracle import torch_oracle as O
    m = synth.load_synth(ConvLSTM((10, 12), 8, [8], [(3, 3)], 1, batch_first=True, bias=True), 21).cuda()
    cell = m.cell_list[0]
    Tn, B = 3, 2
    xs = synth.uniform(21, "x", (B, Tn, 8, 10, 12))
    h0n, c0n = synth.uniform(21, "h0", (B, 8, 10, 12)), synth.uniform(21, "c0", (B, 8, 10, 12))
    ph, pc = synth.uniform(21, "ph", (B, 8, 10, 12)), synth.uniform(21, "pc", (B, 8, 10, 12))
    # oracle
    w = cell.conv.weight.detach().cpu().clone().requires_grad_(True)
    bb = cell.conv.bias.detach().cpu().clone().requires_grad_(True)
    x_c = torch.from_numpy(xs).requires_grad_(True)
    h_c, c_c = torch.from_numpy(h0n).requires_grad_(True), torch.from_numpy(c0n).requires_grad_(True)
    h, c = h_c, c_c
    for t in range(Tn):
        h, c = O.convlstm_cell(w, bb, x_c[:, t], h, c)
    ((h * torch.from_numpy(ph)).sum() + (c * torch.from_numpy(pc)).sum()).backward()
    prev = ops.set_precision(mode)
    try:
        # (a) chained single steps
        x_g = T(xs).requires_grad_(True)
        h_g, c_g = T(h0n).requires_grad_(True), T(c0n).requires_grad_(True)
        hh, cc = h_g, c_g
        for t in range(Tn):
            hh, cc = cell(x_g[:, t], (hh, cc))
        ((hh * T(ph)).sum() + (cc * T(pc)).sum()).backward()
        ga = {"x": x_g.grad.clone(), "h0": h_g.grad.clone(), "c0": c_g.grad.clone(), "w": cell.conv.weight.grad.clone(),
              "b": cell.conv.bias.grad.clone()}
        # (b) the whole sequence with a passed hidden_state
        cell.conv.weight.grad = None; cell.conv.bias.grad = None
        x_g2 = T(xs).requires_grad_(True)
        h_g2, c_g2 = T(h0n).requires_grad_(True), T(c0n).requires_grad_(True)
        out, last = m(x_g2, [(h_g2, c_g2)])
        ((last[0][0] * T(ph)).sum() + (last[0][1] * T(pc)).sum()).backward()
        gb = {"x": x_g2.grad, "h0": h_g2.grad, "c0": c_g2.grad, "w": cell.conv.weight.grad, "b": cell.conv.bias.grad}
    finally:
        ops.set_precision(prev)
    ref = {"x": x_c.grad, "h0": h_c.grad, "c0": c_c.grad, "w": w.grad, "b": bb.grad}

    def rel(a, b):
        return ((a.cpu().double() - b.double()).norm() / b.double().norm()).item()

    for tag, (hv, cv, g) in (("steps", (hh, cc, ga)), ("sequence", (last[0][0], last[0][1], gb))):
        assert rel(hv, h.detach()) <= tol and rel(cv, c.detach()) <= tol, (tag, rel(hv, h.detach()), rel(cv, c.detach()))
        for k in ref:
            r = rel(g[k], ref[k])
            print("%s %-8s %-3s grad rel-L2 %.3e" % (mode, tag, k, r))
            assert r <= tol, (mode, tag, k, r)
    assert out.shape == (B, Tn, 8, 10, 12)


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_packed_images_are_bit_identical_to_the_packing_pass(mode):
    """Packed path (bf16, and split-bf16 with hi + lo planes): producers writing straight into the consumer's packed image
    (ops.PackedImage: encoder chains, the ConvLSTM [x, h] sequence images, h_t only as bf16 for t < T-1 in bf16 mode) must give the
    very numbers of the path that packs every layer's input with jaf_conv2d_pack_input -- outputs, input gradient and every
    parameter gradient.  The backward hand-overs (fused dz, LayerNorm -> convolution) exist in both modes."""
    from jafpro_amd import ops, synth
    from jafpro_amd.crn_model import CRN_smaller
    from jafpro_amd.networks import Accumulate_LSTM_no_loss, UNet_inpainter, VGG19_CRN
    prev = ops.set_precision(mode)
    prev_lazy = ops.set_lazy_resize(False)      # (the fused up-sampling has its own test; it is not bit-identical by design)
    # bf16 storage rides on the packed images (ops.bf16_storage_active) and rounds the stored tensors once more: the identity
    # under test is that of the images themselves, fp32 storage in both arms (tests/test_gpu_kernels.py covers the storage)
    prev_storage = ops.set_bf16_storage(False)
    try:
        for cls, seed, T_ in ((Accumulate_LSTM_no_loss, 21, 3), (UNet_inpainter, 31, 1), (CRN_smaller, 41, 0), (VGG19_CRN, 71, 0)):
            res = []
            for images in (True, False):
                pi = ops.set_packed_images(images)
                try:
                    B = 2
                    if cls is CRN_smaller:
                        m = synth.load_synth(CRN_smaller(3, fg=True), seed).cuda()
                        x = T(synth.uniform(seed, "x", (B, 3, 128, 128))).requires_grad_(True)
                        rgb, mask = m(x, 128)
                        out = torch.cat([rgb, mask], 1)
                    elif cls is VGG19_CRN:
                        m = synth.load_synth(VGG19_CRN(requires_grad=True), seed).cuda()
                        x = T(synth.uniform(seed, "x", (B, 3, 64, 64), -100, 100)).requires_grad_(True)
                        fs = m(x)
                        out = torch.cat([f.reshape(B, -1) for f in fs], 1)
                    else:
                        m = synth.load_synth(cls(), seed).cuda()
                        x = T(synth.uniform(seed, "x", (T_ * B, 72, 200, 200))).requires_grad_(True)
                        out = m.forward_grouped(x, T_) if cls is Accumulate_LSTM_no_loss else m.forward_grouped(x)
                    proj = T(synth.uniform(seed, "proj", tuple(out.shape)))
                    added = ops.SLOT_STATS["added"]
                    handed = ops.FUSED_STATS["dz"]
                    ln_handed = ops.FUSED_STATS["ln"]
                    (out * proj).sum().backward()
                    # with the images on, the last data gradient of a ReLU / LeakyReLU convolution's output hands that layer
                    # its packed dz directly (jaf_packed_io.dz_mask): in the part encoders x1,3,5,7 (from the second of their
                    # two consumers: the ConvLSTM's d x launches / enc_{i+1}) and x2, x4, x6, x8 (one consumer: the stride-2
                    # layer that follows), in VGG the untapped conv -> conv edges;
                    # the reference path packs every dz in a pass of its own
                    # and dec4 (one consumer: the 3-channel output convolution, whose packed input image it writes itself);
                    want = {Accumulate_LSTM_no_loss: 9, UNet_inpainter: 9, VGG19_CRN: 7, CRN_smaller: 0}[cls] if images else 0
                    assert ops.FUSED_STATS["dz"] - handed == want, (cls.__name__, images, ops.FUSED_STATS["dz"] - handed)
                    # every conv -> LayerNorm pair of the CRN gets its dz from the LayerNorm backward (packed bf16, with or without
                    # the images): 13 blocks x 2
                    assert ops.FUSED_STATS["ln"] - ln_handed == (26 if cls is CRN_smaller else 0), ops.FUSED_STATS["ln"] - ln_handed
                    if cls is Accumulate_LSTM_no_loss:
                        # skip features x1, x3, x5, x7 have two consumers (ConvLSTM, enc_{i+1}): with the images on, the
                        # second data gradient is added inside the kernel (ops.GradSlot) -- same numbers, bit for bit
                        assert ops.SLOT_STATS["added"] - added == (4 if images else 0)
                    res.append((out.detach().clone(), x.grad.clone(),
                                {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}))
                finally:
                    ops.set_packed_images(pi)
            (o1, g1, p1), (o0, g0, p0) = res
            assert torch.equal(o1, o0), cls.__name__
            assert torch.equal(g1, g0), cls.__name__
            assert sum(v is not None for v in p0.values()) >= 0.8 * len(p0)
            for k in p0:       # weight gradients: fp32 atomics -> summation order only (None: layers past the last VGG tap)
                assert (p1[k] is None) == (p0[k] is None), k
                if p0[k] is None:
                    continue
                d = (p1[k] - p0[k]).abs().max().item()
                assert d <= 1e-5 * max(1.0, p0[k].abs().max().item()), (cls.__name__, k, d)
    finally:
        ops.set_precision(prev)
        ops.set_lazy_resize(prev_lazy)
        ops.set_bf16_storage(prev_storage)


def test_lazy_resize_matches_resize_then_pack():
    """bf16 path: a decoder's `cat[up(x), skip]` with the up-sampling fused into the packing pass
    (ops.resize(lazy=True) -> jaf_conv2d_pack_input_resized) against the materialised resize: same bilinear taps, so the
    conv output and all gradients agree to bf16 rounding noise of isolated elements (the two kernels may contract the
    interpolation's multiply-adds differently: one fp32 ulp, occasionally one bf16 ulp after rounding)."""
    from jafpro_amd import ops
    G = 3
    # widths below 48 take the gather kernel, the rest the LDS-staged one (ragged tiles, non-integer ratios included)
    for ac, (h, w, OH, OW) in ((True, (13, 13, 25, 25)), (False, (16, 20, 32, 40)), (True, (25, 25, 50, 50)),
                               (False, (50, 50, 100, 100)), (True, (50, 60, 100, 119)), (False, (37, 41, 100, 130))):
        res = []
        for lazy in (True, False):
            g = torch.Generator().manual_seed(5)
            x = (torch.rand(2, G * 8, h, w, generator=g) * 2 - 1).cuda().requires_grad_(True)
            e = (torch.rand(2, 5, h, w, generator=g) * 2 - 1).cuda().requires_grad_(True)       # shared by all groups
            skip = (torch.rand(2, G * 4, OH, OW, generator=g) * 2 - 1).cuda().requires_grad_(True)
            wt = ((torch.rand(G * 6, 17, 3, 3, generator=g) - 0.5) * 0.4).cuda().requires_grad_(True)
            b = (torch.rand(G * 6, generator=g) - 0.5).cuda().requires_grad_(True)
            proj = torch.rand(2, G * 6, OH, OW, generator=g).cuda()
            prev = ops.set_precision("bf16")
            try:
                up = ops.resize(x, (OH, OW), ac, lazy=lazy)
                upe = ops.resize(e, (OH, OW), ac, lazy=lazy)
                assert (getattr(up, "_jaf_lazy", None) is not None) == lazy
                y = ops.conv2d([up, upe, skip], wt, b, stride=1, pad=1, act=ops.ACT_LRELU, slope=0.2, groups=G, shared=[False, True, False])
                (y * proj).sum().backward()
            finally:
                ops.set_precision(prev)
            res.append([t.detach().clone() for t in (y, x.grad, e.grad, skip.grad, wt.grad, b.grad)])
        for a, r, name in zip(res[0], res[1], ("y", "dx", "de", "dskip", "dw", "db")):
            rel = ((a - r).double().norm() / r.double().norm()).item()
            assert rel <= 2e-3, (ac, name, rel)
    # outside the packed bf16 path `lazy` is ignored
    t = torch.rand(1, 2, 4, 4).cuda()
    assert getattr(ops.resize(t, (8, 8), True, lazy=True), "_jaf_lazy", None) is None
