"""GPU parity of every C-ABI kernel against the oracle's torch-CPU fp32 ops (oracle/torch_oracle.py
uses exactly these torch.nn.functional calls).  Tolerances: fp32 MFMA accumulates in a different
order than the CPU GEMM, so conv-like ops are held to 2e-4 * scale; pure elementwise ops to 1e-6."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from jafpro_amd import ops
    return ops


def R(seed, *shape, lo=-1.0, hi=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy(g.uniform(lo, hi, shape).astype(np.float32))


def dev(t):
    return t.cuda().contiguous()


def maxerr(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


CONV_CASES = [
    # N, G, cins, Cout, H, W, k, stride, pad, act
    (2, 1, [5], 7, 19, 23, 3, 1, 1, 1),
    (1, 1, [3], 64, 64, 64, 3, 1, 1, 0),
    (2, 1, [3, 16, 13], 32, 32, 32, 3, 1, 1, 0),       # 3-source concat, odd Cin
    (2, 3, [4], 12, 50, 50, 5, 1, 2, 1),               # grouped 5x5
    (2, 3, [6], 8, 50, 50, 3, 2, 1, 1),                # stride 2, even size
    (1, 2, [8], 16, 25, 25, 3, 2, 1, 1),               # stride 2, odd size 25 -> 13
    (1, 1, [9], 32, 38, 38, 7, 1, 0, 0),               # 7x7 valid
    (2, 1, [32], 3, 30, 37, 7, 1, 3, 0),               # 7x7 padded, two input-channel tiles (weight gradient: kernel rows in two launches)
    (2, 1, [256], 3, 16, 16, 1, 1, 0, 3),              # 1x1 + sigmoid
    (1, 24, [12, 12], 48, 26, 26, 3, 1, 1, 0),         # LSTM-like shape as a plain conv
    (1, 2, [96, 48], 48, 13, 13, 3, 1, 1, 1),
    (1, 1, [6], 32, 256, 256, 3, 2, 1, 1),
    (1, 1, [64], 64, 100, 100, 3, 1, 1, 2),
    (2, 3, [24], 24, 50, 50, 3, 2, 1, 1),              # stride 2, wide (two ci tiles): part-network enc2/4/6/8
    (1, 1, [128], 72, 33, 31, 3, 2, 1, 1),             # stride 2, >32 input channels, odd sizes
    (2, 3, [3], 12, 40, 44, 5, 1, 2, 1),               # 5x5 with <= 8 input channels: paired-tap weight gradient (enc_0)
    (1, 2, [8], 12, 30, 30, 5, 1, 2, 0),
    # wide stride-1 3x3 layers (64 input channels per workgroup: conv_wgrad_fast_kernel).  No activation: with ~10^6 outputs one
    # of them lies within rounding distance of 0, where the LeakyReLU masks of the kernel and of the fp32 CPU reference disagree
    (2, 1, [70, 58], 128, 72, 70, 3, 1, 1, 0),
    (1, 1, [64], 256, 64, 64, 3, 1, 1, 0),
    (1, 2, [40], 48, 66, 50, 3, 1, 1, 0),              # ... 48 and 32 rows per workgroup, two groups, ragged last tile
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_forward_backward(case):
    ops = _ops()
    N, G, cins, Cout, H, W, k, s, p, act = case
    srcs = [R(10 + i, N, G * c, H, W) for i, c in enumerate(cins)]
    Cin = sum(cins)
    w = R(3, G * Cout, Cin, k, k, lo=-0.3, hi=0.3)
    b = R(4, G * Cout)
    # CPU reference: grouped conv over the per-group channel concat
    xs = [t.view(N, G, c, H, W) for t, c in zip(srcs, cins)]
    xcat = torch.cat(xs, 2).reshape(N, G * Cin, H, W).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    y_ref = F.conv2d(xcat, wr, br, stride=s, padding=p, groups=G)
    actf = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.sigmoid}[act]
    y_ref = actf(y_ref)
    proj = R(5, *y_ref.shape)
    (y_ref * proj).sum().backward()

    ds = [dev(t).requires_grad_(True) for t in srcs]
    wd, bd = dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = ops.conv2d(ds, wd, bd, stride=s, pad=p, act=act, slope=0.2, groups=G)
    scale = max(1.0, y_ref.abs().max().item())
    assert y.shape == y_ref.shape
    assert maxerr(y, y_ref) <= 2e-4 * scale
    # the one-thread-per-output kernel must agree as well
    yd = ops.conv2d_direct([t.detach() for t in ds], wd.detach(), bd.detach(), stride=s, pad=p, act=act, slope=0.2, groups=G)
    assert maxerr(yd, y_ref) <= 2e-4 * scale
    (y * dev(proj)).sum().backward()
    gx = xcat.grad.view(N, G, Cin, H, W)
    off = 0
    for t, c in zip(ds, cins):
        ref = gx[:, :, off:off + c].reshape(N, G * c, H, W)
        assert maxerr(t.grad, ref) <= 3e-4 * max(1.0, ref.abs().max().item()), "dgrad source at %d" % off
        off += c
    assert maxerr(wd.grad, wr.grad) <= 3e-4 * max(1.0, wr.grad.abs().max().item()), "wgrad"
    assert maxerr(bd.grad, br.grad) <= 3e-4 * max(1.0, br.grad.abs().max().item()), "bias grad"


def _bf16r(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_bf16_matrix_core_modes(case, mode):
    """The bf16 matrix-core paths (include/jafpro_hip.h JAF_PREC_*).  "bf16": operands are rounded
    to bf16 (RNE) and products accumulate in fp32, so the result must equal a float64 convolution
    of the ROUNDED operands up to fp32 accumulation error (2e-5 * scale).  "bf16x3": hi+lo split,
    must agree with the unrounded fp32 reference to 4e-5 * scale (2^-17 per product)."""
    ops = _ops()
    N, G, cins, Cout, H, W, k, s, p, act = case
    srcs = [R(10 + i, N, G * c, H, W) for i, c in enumerate(cins)]
    Cin = sum(cins)
    w = R(3, G * Cout, Cin, k, k, lo=-0.3, hi=0.3)
    b = R(4, G * Cout)
    rnd = _bf16r if mode == "bf16" else (lambda t: t.double())
    xs = [rnd(t).view(N, G, c, H, W) for t, c in zip(srcs, cins)]
    xcat = torch.cat(xs, 2).reshape(N, G * Cin, H, W).requires_grad_(True)
    wr = rnd(w).requires_grad_(True)
    z_ref = F.conv2d(xcat, wr, b.double(), stride=s, padding=p, groups=G)
    actf = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu, 3: torch.sigmoid}[act]
    y_ref = actf(z_ref)
    proj = R(5, *y_ref.shape)
    # dgrad reference: the kernel rounds dz (not dy) to bf16
    dz = torch.autograd.grad(y_ref, z_ref, proj.double(), retain_graph=True)[0]
    gx, gw = torch.autograd.grad(z_ref, [xcat, wr], rnd(dz.float()))
    gx = gx.view(N, G, Cin, H, W)
    prev = ops.set_precision(mode)
    try:
        ds = [dev(t).requires_grad_(True) for t in srcs]
        wd, bd = dev(w).requires_grad_(True), dev(b).requires_grad_(True)
        y = ops.conv2d(ds, wd, bd, stride=s, pad=p, act=act, slope=0.2, groups=G)
        (y * dev(proj)).sum().backward()
    finally:
        ops.set_precision(prev)
    tol = 2e-5 if mode == "bf16" else 4e-5
    scale = max(1.0, z_ref.abs().max().item())
    assert maxerr(y, y_ref) <= tol * scale
    off = 0
    for t, c in zip(ds, cins):
        ref = gx[:, :, off:off + c].reshape(N, G * c, H, W)
        # dz is rounded to bf16 on the GPU from an fp32 value that differs from the float64
        # reference in the last bits: a handful of round-to-nearest ties flip (one bf16 ulp of
        # one dz element); everything else must be at accumulation-order accuracy.
        err = (t.grad.detach().cpu().double() - ref).abs()
        sc = max(1.0, ref.abs().max().item())
        assert (err > 2 * tol * sc).double().mean().item() <= 2e-3, "dgrad source at %d" % off
        assert err.max().item() <= 8e-3 * sc, "dgrad source at %d" % off
        off += c
    # weight gradient: bf16 matrix cores for 1x1 / 3x3 / 5x5 / 7x7 layers in "bf16" mode (x and dz rounded), the
    # exact fp32 kernel otherwise -- either way it must match the float64 reference of what was rounded.
    if mode == "bf16" and k not in (1, 3, 5, 7):
        gw = torch.autograd.grad(F.conv2d(torch.cat([t.double().view(N, G, c, H, W) for t, c in zip(srcs, cins)], 2)
                                          .reshape(N, G * Cin, H, W), wr, None, stride=s, padding=p, groups=G),
                                 wr, dz)[0]
    err = (wd.grad.detach().cpu().double() - gw).abs()
    sc = max(1.0, gw.abs().max().item())
    assert (err > 4 * tol * sc).double().mean().item() <= 2e-3, "wgrad"
    assert err.max().item() <= 8e-3 * sc, "wgrad"


def test_conv2d_shared_source():
    """inpainter dec1: the 72-channel global embed is read by all 24 groups (src/networks.py:1164)."""
    ops = _ops()
    N, G, H = 2, 4, 13
    a, e, c = R(1, N, G * 6, H, H), R(2, N, 5, H, H), R(3, N, G * 3, H, H)
    w, b = R(4, G * 8, 14, 3, 3, lo=-0.3, hi=0.3), R(5, G * 8)
    ar, er, cr = a.clone().requires_grad_(True), e.clone().requires_grad_(True), c.clone().requires_grad_(True)
    outs = []
    for g in range(G):
        xin = torch.cat([ar[:, g * 6:(g + 1) * 6], er, cr[:, g * 3:(g + 1) * 3]], 1)
        outs.append(F.leaky_relu(F.conv2d(xin, w[g * 8:(g + 1) * 8], b[g * 8:(g + 1) * 8], padding=1), 0.2))
    y_ref = torch.cat(outs, 1)
    proj = R(6, *y_ref.shape)
    (y_ref * proj).sum().backward()
    ad, ed, cd = (dev(t).requires_grad_(True) for t in (a, e, c))
    y = ops.conv2d([ad, ed, cd], dev(w), dev(b), stride=1, pad=1, act=1, slope=0.2, groups=G, shared=[False, True, False])
    assert maxerr(y, y_ref) <= 2e-4 * max(1.0, y_ref.abs().max().item())
    (y * dev(proj)).sum().backward()
    for t, r in ((ad, ar), (ed, er), (cd, cr)):
        assert maxerr(t.grad, r.grad) <= 3e-4 * max(1.0, r.grad.abs().max().item())


@pytest.mark.parametrize("mode,tol", [("bf16x3", 2e-4), ("bf16", 3e-2)])
def test_convlstm_sequence_matrix_core_modes(mode, tol):
    """ConvLSTM on the bf16 matrix-core paths vs the fp32 oracle (h_T and the input gradient)."""
    from oracle import torch_oracle as O
    ops = _ops()
    T, B, G, C, H, W = 3, 2, 2, 12, 20, 20
    x = R(1, T, B, G * C, H, W)
    w = R(2, G * 4 * C, 2 * C, 3, 3, lo=-0.3, hi=0.3)
    b = R(3, G * 4 * C, lo=-0.2, hi=0.2)
    xr = x.clone().requires_grad_(True)
    hT = []
    for g in range(G):
        seq = [xr[t][:, g * C:(g + 1) * C] for t in range(T)]
        _, (h, c) = O.convlstm(w[g * 4 * C:(g + 1) * 4 * C], b[g * 4 * C:(g + 1) * 4 * C], seq)
        hT.append(h)
    h_ref = torch.cat(hT, 1)
    proj = R(4, *h_ref.shape)
    (h_ref * proj).sum().backward()
    prev = ops.set_precision(mode)
    try:
        xd = dev(x).requires_grad_(True)
        h, _ = ops.convlstm(xd, dev(w).requires_grad_(True), dev(b).requires_grad_(True), groups=G)
        (h * dev(proj)).sum().backward()
    finally:
        ops.set_precision(prev)
    assert maxerr(h, h_ref) <= tol
    assert maxerr(xd.grad, xr.grad) <= 4 * tol * max(1.0, xr.grad.abs().max().item())


@pytest.mark.parametrize("shape", [(3, 2, 1, 4, 7, 5), (4, 1, 3, 12, 20, 20), (2, 2, 2, 24, 13, 13)])
def test_convlstm_sequence(shape):
    from oracle import torch_oracle as O
    ops = _ops()
    T, B, G, C, H, W = shape
    x = R(1, T, B, G * C, H, W)
    w = R(2, G * 4 * C, 2 * C, 3, 3, lo=-0.3, hi=0.3)
    b = R(3, G * 4 * C, lo=-0.2, hi=0.2)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    hT = []
    for g in range(G):
        seq = [xr[t][:, g * C:(g + 1) * C] for t in range(T)]
        _, (h, c) = O.convlstm(wr[g * 4 * C:(g + 1) * 4 * C], br[g * 4 * C:(g + 1) * 4 * C], seq)
        hT.append(h)
    h_ref = torch.cat(hT, 1)
    proj = R(4, *h_ref.shape)
    (h_ref * proj).sum().backward()
    xd, wd, bd = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    h, c_last = ops.convlstm(xd, wd, bd, groups=G, return_all=False)
    assert maxerr(h, h_ref) <= 1e-4
    (h * dev(proj)).sum().backward()
    assert maxerr(xd.grad, xr.grad) <= 2e-4 * max(1.0, xr.grad.abs().max().item())
    assert maxerr(wd.grad, wr.grad) <= 3e-4 * max(1.0, wr.grad.abs().max().item())
    assert maxerr(bd.grad, br.grad) <= 3e-4 * max(1.0, br.grad.abs().max().item())


def test_layernorm_lrelu():
    from oracle import torch_oracle as O
    ops = _ops()
    x, g, b = R(1, 3, 10, 17, 9, lo=-2, hi=3), R(2, 10, lo=0.2, hi=1.0), R(3, 10, lo=-0.2, hi=0.2)
    xr, gr, br = (t.clone().requires_grad_(True) for t in (x, g, b))
    y_ref = F.leaky_relu(O.crn_layernorm(xr, gr, br), 0.01)
    proj = R(4, *x.shape)
    (y_ref * proj).sum().backward()
    xd, gd, bd = (dev(t).requires_grad_(True) for t in (x, g, b))
    y = ops.layernorm_lrelu(xd, gd, bd, 1e-5, 0.01)
    assert maxerr(y, y_ref) <= 2e-6 * max(1.0, y_ref.abs().max().item())
    (y * dev(proj)).sum().backward()
    assert maxerr(xd.grad, xr.grad) <= 1e-5
    assert maxerr(gd.grad, gr.grad) <= 1e-4
    assert maxerr(bd.grad, br.grad) <= 1e-4


@pytest.mark.parametrize("N,Cin,C,H,W", [(2, 8, 16, 12, 20), (3, 5, 10, 9, 7), (1, 16, 24, 32, 32)])
def test_layernorm_backward_hands_packed_dz_to_its_convolution(N, Cin, C, H, W):
    """bf16 path, conv (no activation) -> LayerNorm -> LeakyReLU with the LayerNorm as the convolution's only reader
    (src/crn_model.py:90-106): the LayerNorm backward writes the convolution's packed bf16 dz and its bias gradient
    (jaf_layernorm_lrelu_bwd_packed) instead of an fp32 dx that jaf_conv2d_pack_dz would read back.  Same expressions in
    both paths: the data and weight gradients agree to a bf16 ulp of isolated elements, the bias gradient (summed
    analytically instead of element by element) to fp32 rounding."""
    ops = _ops()
    prev = ops.set_precision("bf16")
    try:
        x, w, b = R(1, N, Cin, H, W), R(2, C, Cin, 3, 3, lo=-0.3, hi=0.3), R(3, C, lo=-0.2, hi=0.2)
        g, be = R(4, C, lo=0.3, hi=1.2), R(5, C, lo=-0.2, hi=0.2)
        proj = dev(R(6, N, C, H, W))
        res = []
        for sole in (True, False):
            xd, wd, bd, gd, bed = (dev(t).requires_grad_(True) for t in (x, w, b, g, be))
            handed = ops.FUSED_STATS["ln"]
            y = ops.conv2d(xd, wd, bd, stride=1, pad=1, act=0)
            z = ops.layernorm_lrelu(y, gd, bed, 1e-5, 0.01, sole_consumer=sole)
            (z * proj).sum().backward()
            assert ops.FUSED_STATS["ln"] - handed == (1 if sole else 0)
            res.append((z.detach().clone(), xd.grad, wd.grad, bd.grad, gd.grad, bed.grad))
        a, r = res
        assert torch.equal(a[0], r[0])
        for k, name in ((1, "dx"), (2, "dw")):
            scale = r[k].abs().max().item()
            d = (a[k] - r[k]).abs()
            assert d.max().item() <= 2e-2 * scale, (name, d.max().item(), scale)               # a flipped bf16 ulp of one dz element
            assert (d > 1e-6 * scale).float().mean().item() <= 2e-2, (name, (d > 1e-6 * scale).float().mean().item())
        assert maxerr(a[3], r[3].cpu()) <= 2e-5 * max(1.0, r[3].abs().max().item()), "conv bias gradient"
        for k in (4, 5):         # gamma / beta: the same reduce pass (fp32 atomics over the images: summation order only)
            assert maxerr(a[k], r[k].cpu()) <= 1e-5 * max(1.0, r[k].abs().max().item())
    finally:
        ops.set_precision(prev)


@pytest.mark.parametrize("shape", [(3, 6, 11, 7), (2, 8, 16, 12), (2, 4, 200, 180)],
                         ids=["one-launch-scalar", "one-launch-16B", "three-kernel"])
@pytest.mark.parametrize("act,res,training", [(2, False, True), (0, True, True), (1, False, True), (2, False, False)])
def test_batchnorm_act(act, res, training, shape):
    """(feature maps of at most 65536 elements per channel take the statistics and apply them in one launch, forward and
    backward: jaf_batchnorm_act_fwd_fused / jaf_batchnorm_act_bwd; the third shape runs the multi-kernel path)"""
    ops = _ops()
    N, C, H, W = shape
    x, w, b = R(1, N, C, H, W, lo=-2, hi=2), R(2, C, lo=0.5, hi=1.5), R(3, C, lo=-0.3, hi=0.3)
    rm, rv = R(4, C, lo=-0.1, hi=0.1), R(5, C, lo=0.8, hi=1.2)
    rsd = R(6, N, C, H, W)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    rr = rsd.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    y_ref = F.batch_norm(xr, rm_r, rv_r, wr, br, training, 0.1, 1e-5)
    y_ref = {0: lambda t: t, 1: lambda t: F.leaky_relu(t, 0.2), 2: F.relu}[act](y_ref)
    if res:
        y_ref = y_ref + rr
    proj = R(7, N, C, H, W)
    (y_ref * proj).sum().backward()
    xd, wd, bd = (dev(t).requires_grad_(True) for t in (x, w, b))
    rd = dev(rsd).requires_grad_(True)
    rm_d, rv_d = dev(rm), dev(rv)
    y = ops.batchnorm_act(xd, wd, bd, rm_d, rv_d, training, act, 0.2, rd if res else None)
    assert maxerr(y, y_ref) <= 5e-6 * max(1.0, y_ref.abs().max().item())
    assert maxerr(rm_d, rm_r) <= 1e-6 and maxerr(rv_d, rv_r) <= 1e-6
    (y * dev(proj)).sum().backward()
    sc = lambda t: max(1.0, t.abs().max().item())
    assert maxerr(xd.grad, xr.grad) <= 2e-5 * sc(xr.grad)
    assert maxerr(wd.grad, wr.grad) <= 1e-4 * sc(wr.grad) and maxerr(bd.grad, br.grad) <= 1e-4 * sc(br.grad)
    if res:
        assert maxerr(rd.grad, rr.grad) <= 1e-6


@pytest.mark.parametrize("k,s,p,H,W", [(3, 2, 1, 16, 16), (3, 2, 1, 9, 7), (2, 2, 0, 8, 12), (2, 2, 0, 10, 24), (3, 2, 1, 12, 20)])
def test_avgpool(k, s, p, H, W):
    ops = _ops()
    x = R(1, 2, 5, H, W)
    xr = x.clone().requires_grad_(True)
    y_ref = F.avg_pool2d(xr, k, s, p)
    proj = R(2, *y_ref.shape)
    (y_ref * proj).sum().backward()
    xd = dev(x).requires_grad_(True)
    y = ops.avg_pool(xd, k, s, p)
    assert maxerr(y, y_ref) <= 1e-6
    (y * dev(proj)).sum().backward()
    assert maxerr(xd.grad, xr.grad) <= 1e-6


@pytest.mark.parametrize("H,W,OH,OW,ac", [(13, 13, 25, 25, True), (25, 25, 50, 50, True), (256, 256, 4, 4, True),
                                           (8, 8, 16, 16, True), (16, 16, 32, 32, False), (64, 64, 13, 9, False)])
def test_resize_bilinear(H, W, OH, OW, ac):
    ops = _ops()
    x = R(1, 2, 3, H, W)
    xr = x.clone().requires_grad_(True)
    y_ref = F.interpolate(xr, size=(OH, OW), mode="bilinear", align_corners=ac)
    proj = R(2, *y_ref.shape)
    (y_ref * proj).sum().backward()
    xd = dev(x).requires_grad_(True)
    y = ops.resize(xd, (OH, OW), ac)
    assert maxerr(y, y_ref) <= 2e-6
    (y * dev(proj)).sum().backward()
    assert maxerr(xd.grad, xr.grad) <= 1e-5


@pytest.mark.parametrize("C,H,W,OH,OW,ac", [(2400, 50, 50, 100, 100, True), (2400, 37, 41, 74, 82, False), (1200, 64, 48, 100, 100, True)])
def test_resize_adjoint_many_planes(C, H, W, OH, OW, ac):
    """Launches with >= 2048 workgroups take the tall-tile form of the adjoint kernel (several input rows per lane, all of a
    lane's staging loads in flight at once): against ATen's bilinear backward."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(1, C, H, W, generator=g) - 0.5).cuda().requires_grad_(True)
    proj = (torch.rand(1, C, OH, OW, generator=g) - 0.5).cuda()
    ops.resize(x, (OH, OW), ac).mul(proj).sum().backward()
    xr = x.detach().cpu().double().requires_grad_(True)
    (F.interpolate(xr, size=(OH, OW), mode="bilinear", align_corners=ac) * proj.cpu().double()).sum().backward()
    assert maxerr(x.grad, xr.grad) <= 1e-5


def test_resize_crop_and_nearest():
    """face crops, train/4...py:342-350: bilinear (AC=False) and nearest to 64x64 of a box."""
    ops = _ops()
    x = R(1, 2, 3, 256, 256)
    y0, y1, x0, x1 = 32, 96, 96, 160
    crop = x[:, :, y0:y1, x0:x1]
    ref_b = F.interpolate(crop, size=(64, 64), mode="bilinear", align_corners=False)
    ref_n = F.interpolate(x[:, :, 30:77, 90:141], size=(64, 64), mode="nearest")
    xd = dev(x)
    assert maxerr(ops.resize(xd, (64, 64), False, crop=(y0, x0, y1 - y0, x1 - x0)), ref_b) <= 2e-6
    assert maxerr(ops.resize(xd, (64, 64), False, nearest=True, crop=(30, 90, 47, 51)), ref_n) == 0.0


@pytest.mark.parametrize("p", [1, 3])
def test_reflect_pad(p):
    ops = _ops()
    x = R(1, 2, 3, 9, 12)
    xr = x.clone().requires_grad_(True)
    y_ref = F.pad(xr, (p, p, p, p), mode="reflect")
    proj = R(2, *y_ref.shape)
    (y_ref * proj).sum().backward()
    xd = dev(x).requires_grad_(True)
    y = ops.reflect_pad(xd, p)
    assert maxerr(y, y_ref) == 0.0
    (y * dev(proj)).sum().backward()
    assert maxerr(xd.grad, xr.grad) <= 1e-6


@pytest.mark.parametrize("ac", [False, True])
def test_texture_warp(ac):
    from oracle import torch_oracle as O
    from jafpro_amd import synth
    ops = _ops()
    B = 2
    iuv = synth.iuv255(5, "iuv", B, 256)
    iuv[0, 0, 0] = (3, 0, 255)          # corner taps: exercises the zero padding
    iuv[0, 0, 1] = (3, 255, 0)
    tex = R(1, B, 72, 200, 200)
    tr = tex.clone().requires_grad_(True)
    ref = torch.stack([O.texture_warp([tr[b, 3 * p:3 * p + 3] for p in range(24)], iuv[b], ac) for b in range(B)])
    proj = R(2, *ref.shape)
    (ref * proj).sum().backward()
    td = dev(tex).requires_grad_(True)
    out = ops.texture_warp(td, torch.from_numpy(iuv).cuda(), ac)
    assert maxerr(out, ref) <= 2e-6
    (out * dev(proj)).sum().backward()
    assert maxerr(td.grad, tr.grad) <= 2e-5


@pytest.mark.parametrize("border,ac", [(True, False), (False, False), (True, True)])
def test_grid_sample(border, ac):
    ops = _ops()
    src = R(1, 2, 3, 40, 37)
    grid = R(2, 2, 21, 19, 2, lo=-1.3, hi=1.3)
    grid[0, 0, 0] = -2.0
    ref = F.grid_sample(src, grid, mode="bilinear", padding_mode="border" if border else "zeros", align_corners=ac)
    out = ops.grid_sample(dev(src), dev(grid), border, ac)
    assert maxerr(out, ref) <= 2e-6


def test_blend_and_mul():
    ops = _ops()
    a, b, m = R(1, 2, 3, 16, 16), R(2, 2, 3, 16, 16), R(3, 2, 1, 16, 16, lo=0, hi=1)
    ar, br, mr = (t.clone().requires_grad_(True) for t in (a, b, m))
    ref = ar * mr.repeat(1, 3, 1, 1) + br * (1 - mr.repeat(1, 3, 1, 1))
    proj = R(4, *ref.shape)
    (ref * proj).sum().backward()
    ad, bd, md = (dev(t).requires_grad_(True) for t in (a, b, m))
    out = ops.blend(ad, bd, md)
    assert maxerr(out, ref) <= 1e-6
    (out * dev(proj)).sum().backward()
    for t, r in ((ad, ar), (bd, br), (md, mr)):
        assert maxerr(t.grad, r.grad) <= 1e-6
    m3 = R(5, 2, 3, 16, 16, lo=0, hi=1)
    assert maxerr(ops.mul_bcast(dev(a), dev(m3)), a * m3) <= 1e-7
    assert maxerr(ops.mul_bcast(dev(a), dev(m)), a * m) <= 1e-7


def test_part_mask_and_atlas():
    from oracle import torch_oracle as O
    from jafpro_amd import synth
    ops = _ops()
    B, T = 2, 4
    atlas = R(1, B, T, 3, 800, 1200)
    parts = ops.atlas_to_parts(dev(atlas))
    ref = torch.cat([torch.cat([atlas[:, t, :, i * 200:(i + 1) * 200, j * 200:(j + 1) * 200]
                                for i in range(4) for j in range(6)], 1) for t in range(T)], 0)
    assert maxerr(parts, ref) == 0.0
    masks = torch.from_numpy(synth.rect_masks(3, "m", (B, T, 800, 1200)))
    tex = R(2, B, 72, 200, 200)
    used = [0, 2]
    area = O.common_area_mask(masks, used)
    ref = torch.cat(O.mask_parts([tex[:, 3 * p:3 * p + 3] for p in range(24)], area), 1)
    u = torch.zeros(T, dtype=torch.int32); u[used] = 1
    out = ops.part_mask_mul(dev(tex), dev(masks), u.cuda())
    assert maxerr(out, ref) == 0.0


def test_losses_linear_adam():
    ops = _ops()
    a, b = R(1, 2, 3, 33, 17), R(2, 2, 3, 33, 17)
    ar = a.clone().requires_grad_(True)
    ref = 0.7 * F.l1_loss(ar, b)
    ref.backward()
    ad = dev(a).requires_grad_(True)
    out = ops.l1_loss(ad, dev(b), 0.7)
    assert abs(out.item() - ref.item()) <= 1e-6
    out.sum().backward()
    assert maxerr(ad.grad, ar.grad) <= 1e-9
    # vgg preprocess
    from oracle import torch_oracle as O
    assert maxerr(ops.vgg_preprocess(dev(a)), O.vgg_preprocess(a)) <= 2e-5
    # BCE
    p = R(3, 5, 1, lo=0.01, hi=0.99)
    for tgt in (0.0, 1.0):
        pr = p.clone().requires_grad_(True)
        ref = F.binary_cross_entropy(pr, torch.full_like(pr, tgt))
        ref.backward()
        pd = dev(p).requires_grad_(True)
        out = ops.bce_loss(pd, tgt)
        assert abs(out.item() - ref.item()) <= 1e-6
        out.sum().backward()
        assert maxerr(pd.grad, pr.grad) <= 1e-6
    # linear
    x, w, bb = R(4, 3, 70), R(5, 11, 70, lo=-0.2, hi=0.2), R(6, 11)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, bb))
    ref = F.leaky_relu(F.linear(xr, wr, br), 0.2)
    proj = R(7, 3, 11)
    (ref * proj).sum().backward()
    xd, wd, bd = (dev(t).requires_grad_(True) for t in (x, w, bb))
    out = ops.linear(xd, wd, bd, 1, 0.2)
    assert maxerr(out, ref) <= 1e-5
    (out * dev(proj)).sum().backward()
    for t, r in ((xd, xr), (wd, wr), (bd, br)):
        assert maxerr(t.grad, r.grad) <= 1e-5
    # Adam vs torch.optim.Adam, 3 steps
    pp = R(8, 1003)
    pr = pp.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    pd, m, v = dev(pp), torch.zeros(1003).cuda(), torch.zeros(1003).cuda()
    for step in range(1, 4):
        g = R(20 + step, 1003)
        pr.grad = g.clone()
        opt.step()
        ops.adam_step(pd, dev(g), m, v, 1e-3, step)
    assert maxerr(pd, pr) <= 1e-6


def test_rasterizer_and_flow():
    """fim must be bit-exact vs the C restatement; weights/flow within 1e-6."""
    from oracle import raster_oracle, torch_oracle as O
    from jafpro_amd import synth
    ops = _ops()
    B = 2
    vs, vt = synth.posed_vertices(81, "src", B), synth.posed_vertices(81, "tgt", B)
    cam = np.zeros((B, 3), np.float32); cam[:, 0] = 0.9
    _, fidx = synth.body_mesh()
    f_ref = O.project_faces(torch.from_numpy(vt), torch.from_numpy(cam), fidx)
    f_gpu = ops.project_faces(torch.from_numpy(vt).cuda(), torch.from_numpy(cam).cuda(), torch.from_numpy(fidx).cuda(), float(np.float32(O.EYE_Z)))
    assert maxerr(f_gpu, f_ref) == 0.0
    fim_ref, wim_ref = raster_oracle.rasterize_fim_wim(f_ref.numpy(), 256)
    fim, wim = ops.rasterize_fim_wim(f_gpu, 256)
    assert (fim.cpu().numpy() == fim_ref).all(), "face index map differs at %d pixels" % (fim.cpu().numpy() != fim_ref).sum()
    assert np.abs(wim.cpu().numpy() - wim_ref).max() == 0.0
    fs_ref = O.project_faces(torch.from_numpy(vs), torch.from_numpy(cam), fidx)
    img = R(1, B, 3, 256, 256)
    warped_ref, T_ref = O.flow_warp(img, fs_ref, torch.from_numpy(fim_ref), torch.from_numpy(wim_ref))
    T_gpu = ops.bc_transform(fs_ref.cuda().contiguous(), fim, wim)
    assert maxerr(T_gpu, T_ref) <= 1e-6
    assert maxerr(ops.grid_sample(dev(img), T_gpu, True, False), warped_ref) <= 1e-5


def test_rasterizer_small_cases():
    """single triangle, degenerate and back-facing faces, batch index (tests/utils.py:11-27 idea)."""
    from oracle import raster_oracle
    ops = _ops()
    faces = np.zeros((3, 4, 3, 3), np.float32)
    faces[1, 0] = [[-0.5, -0.5, 2.0], [0.5, -0.5, 2.0], [0.0, 0.6, 2.5]]          # front (ccw)
    faces[1, 1] = [[-0.5, -0.5, 1.5], [0.0, 0.6, 1.5], [0.5, -0.5, 1.5]]          # back-facing, nearer: culled
    faces[1, 2] = [[-0.2, -0.2, 1.0], [0.3, -0.2, 1.0], [0.0, 0.3, 1.0]]          # front, nearest
    faces[1, 3] = [[-0.2, -0.2, 1.0], [0.3, -0.2, 1.0], [0.0, 0.3, 1.0]]          # duplicate: lower index must win
    faces[2, 0] = [[-0.9, -0.9, 150.0], [0.9, -0.9, 150.0], [0.0, 0.9, 150.0]]    # beyond far
    for S in (64, 256):
        fr, wr = raster_oracle.rasterize_fim_wim(faces, S)
        fg, wg = ops.rasterize_fim_wim(torch.from_numpy(faces).cuda(), S)
        assert (fg.cpu().numpy() == fr).all()
        assert np.abs(wg.cpu().numpy() - wr).max() == 0.0
        assert (fr[0] == -1).all() and (fr[2] == -1).all() and (fr[1] == 2).any() and not (fr[1] == 3).any()


@pytest.mark.parametrize("T,N,G,C,H,W", [(2, 2, 4, 12, 72, 40), (2, 2, 3, 24, 100, 100), (1, 2, 3, 12, 50, 50), (2, 1, 2, 12, 64, 48)])
def test_convlstm_h_in_packed_image(T, N, G, C, H, W):
    """bf16 path: the cell kernel writes h_T straight into a channel slot of the consumer's packed image
    (jaf_packed_io.dst).  With 4 pixels per lane (widths that are multiples of the 16-pixel tile rows) the four row groups of
    a wave exchange their values with v_permlane{32,16}_swap so that each lane stores 4 channels of one pixel: the image must
    hold exactly RNE-bf16(h), nothing outside the slot may be touched.  (72 x 40: ragged tiles, dead lanes take part in the
    exchange; 50 x 50 and 64 x 48: the scalar store path.)"""
    ops = _ops()
    prev = ops.set_precision("bf16")
    try:
        x = dev(R(1, T, N, G * C, H, W, lo=-1, hi=1))
        w = dev(R(2, G * 4 * C, 2 * C, 3, 3, lo=-0.15, hi=0.15))
        b = dev(R(3, G * 4 * C, lo=-0.1, hi=0.1))
        img = ops.PackedImage(2 * N, G, 2 * C, H, W, "cuda", zero=True)
        with torch.no_grad():
            h, c = ops.convlstm(x, w, b, groups=G, final_dst=img.slot(C, N))
        buf = img.buf.view(torch.bfloat16).float().reshape(2 * N, G, img.ng8, H * W, 8)
        full = buf.permute(0, 1, 2, 4, 3).reshape(2 * N, G, img.ng8 * 8, H, W)
        ref = h.reshape(N, G, C, H, W).to(torch.bfloat16).float()
        assert torch.equal(full[N:, :, C:2 * C], ref)
        assert full[:N].abs().max().item() == 0.0 and full[N:, :, :C].abs().max().item() == 0.0
        assert full[N:, :, 2 * C:].numel() == 0 or full[N:, :, 2 * C:].abs().max().item() == 0.0
    finally:
        ops.set_precision(prev)


@pytest.mark.parametrize("C,H,W,g16,first", [(12, 20, 20, True, False), (12, 10, 14, True, True), (24, 10, 10, True, False),
                                             (24, 6, 6, False, False), (48, 5, 5, True, False), (12, 5, 5, True, False),
                                             (4, 6, 6, True, False), (12, 20, 20, False, False)])
def test_lstm_gates_bwd_packed_direct(C, H, W, g16, first):
    """jaf_convlstm_gates_bwd_packed (src/convLSTM.py:48-54 adjoint) against the closed form: dc_prev,
    the packed bf16 pre-activation gate gradients [n][g][4C/8][px][8] (channel 4 c + gate) and the bias gradient.  Covers 4,
    2 and 1 pixels per lane (H*W % 4 == 0, even, odd)."""
    import ctypes
    ops = _ops()
    from jafpro_amd._lib import lib
    N, G = 2, 3
    HW = H * W
    gates = torch.cat([torch.sigmoid(R(1, N, G, 3, C, H, W, lo=-2, hi=2)), torch.tanh(R(2, N, G, 1, C, H, W, lo=-2, hi=2))], 2)
    if g16:
        gates = gates.to(torch.bfloat16)
    gf = gates.float()
    dh, dcn, cp, cc = (R(s, N, G, C, H, W, lo=-1, hi=1) for s in (3, 4, 5, 6))
    i, f, o, gg = gf[:, :, 0], gf[:, :, 1], gf[:, :, 2], gf[:, :, 3]
    tc = torch.tanh(cc)
    dc = dh * o * (1 - tc * tc) + dcn
    cpv = torch.zeros_like(cp) if first else cp
    dgates = torch.stack([dc * gg * i * (1 - i), dc * cpv * f * (1 - f), dh * tc * o * (1 - o), dc * i * (1 - gg * gg)], 2)
    dcp_ref = dc * f
    db_ref = dgates.sum((0, 4, 5)).reshape(-1)
    d = {k: dev(v.contiguous()) for k, v in dict(dh=dh, dcn=dcn, cp=cp, cc=cc).items()}
    # fp32 gates: planes [n][g][gate][c][px]; bf16 gates (what the packed cell kernel saves): gate-innermost [n][g][c][px][gate]
    gd = (gates.permute(0, 1, 3, 4, 5, 2) if g16 else gates.reshape(N, G * 4 * C, H, W)).contiguous().cuda()
    dcp = torch.empty(N, G * C, H, W, device="cuda")
    ng8 = 4 * C // 8
    packed = torch.zeros(N * G * ng8 * HW * 8, device="cuda", dtype=torch.bfloat16)
    db = torch.zeros(G * 4 * C, device="cuda")
    rc = lib().jaf_convlstm_gates_bwd_packed(ops._s(), N, G, C, HW, ops._p(d["dh"]), ops._p(d["dcn"]), ops._p(gd), 1 if g16 else 0,
                                            None if first else ops._p(d["cp"]), ops._p(d["cc"]), ops._p(dcp), ops._p(packed), ops._p(db))
    assert rc == 0
    torch.cuda.synchronize()
    assert maxerr(dcp.reshape(N, G, C, H, W), dcp_ref) <= 2e-6
    # packed channels are channel-major: 4 * c + gate (one 16-byte item = 2 hidden channels x i, f, o, g)
    got = packed.float().cpu().reshape(N, G, ng8, HW, 8).permute(0, 1, 2, 4, 3).reshape(N, G, C, 4, H, W).permute(0, 1, 3, 2, 4, 5)
    # bf16 rounding of values up to ~1: half an ulp is 2^-9 relative
    assert (got - dgates).abs().max().item() <= 2.0 ** -8 * max(1.0, dgates.abs().max().item())
    assert maxerr(db, db_ref) <= 1e-4 * max(1.0, db_ref.abs().max().item())


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 64, 32, 32), (3, 7, 20, 13, 9), (1, 64, 256, 16, 16)])
def test_layernorm_statistics_from_conv_epilogue(N, Cin, Cout, H, W):
    """jaf_conv2d_fwd_packed_stats + jaf_layernorm_finalize (bf16 packed path): the LayerNorm fed by the sums the
    convolution's epilogue accumulated must equal the LayerNorm that makes its own statistics pass over
    the same convolution output (src/crn_model.py:78-87)."""
    ops = _ops()
    x, w, b = dev(R(1, N, Cin, H, W)), dev(R(2, Cout, Cin, 3, 3, lo=-0.2, hi=0.2)), dev(R(3, Cout, lo=-0.5, hi=0.5))
    g, be = dev(R(4, Cout, lo=0.2, hi=1.0)), dev(R(5, Cout, lo=-0.2, hi=0.2))
    prev = ops.set_precision("bf16")
    try:
        st = ops.LNStats()
        y = ops.conv2d(x, w, b, stride=1, pad=1, act=0, ln_stats=st)
        assert st.filled and st.sums.numel() == N * st.slots * 2
        sums = st.sums.view(N, st.slots, 2).sum(1).cpu()
        fused = ops.layernorm_lrelu(y, g, be, 1e-5, 0.01, st)
        assert not st.filled and float(st.sums.abs().max()) == 0.0      # consumed, and left clean for the next use
        plain = ops.layernorm_lrelu(y, g, be, 1e-5, 0.01)
        # second round through the same side channel: no re-zeroing by the caller
        y2 = ops.conv2d(x * 0.5, w, b, stride=1, pad=1, act=0, ln_stats=st)
        fused2 = ops.layernorm_lrelu(y2, g, be, 1e-5, 0.01, st)
        plain2 = ops.layernorm_lrelu(y2, g, be, 1e-5, 0.01)
        assert maxerr(fused2, plain2.cpu()) <= 2e-5
    finally:
        ops.set_precision(prev)
    ref = torch.stack([y.double().sum((1, 2, 3)), (y.double() ** 2).sum((1, 2, 3))], 1).cpu()
    assert torch.allclose(sums, ref, rtol=2e-6, atol=1e-6)
    assert maxerr(fused, plain.cpu()) <= 2e-5
    # other paths leave the side channel untouched
    st2 = ops.LNStats()
    ops.conv2d(x, w, b, stride=1, pad=1, act=0, ln_stats=st2)      # f32 mode
    assert not st2.filled


def test_launches_split_over_the_batch_beyond_the_grid_z_extent(monkeypatch):
    """The per-plane kernels put (image, channel) on blockIdx.z (<= 65535): larger batches go out as several launches over
    image ranges (ops._n_chunks).  With the limit lowered to 7 planes the chunked results must equal the single launch."""
    ops = _ops()
    x = dev(R(1, 5, 6, 20, 24))
    w, b = dev(R(2, 8, 6, 3, 3)), dev(R(3, 8))

    def run():
        xr = x.clone().requires_grad_(True)
        wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.avg_pool(ops.reflect_pad(ops.resize(xr, (30, 36), True), 2), 3, 2, 1)
        z = ops.conv2d(y, wr, br, stride=1, pad=1, act=ops.ACT_LRELU, slope=0.2)
        (z * z).sum().backward()
        return z.detach(), xr.grad, wr.grad, br.grad

    for mode in ("f32", "bf16"):
        prev = ops.set_precision(mode)
        try:
            ref = run()
            monkeypatch.setattr(ops, "_GRID_Z", 7)
            got = run()
            monkeypatch.setattr(ops, "_GRID_Z", 65535)
        finally:
            ops.set_precision(prev)
        assert len(ops._n_chunks(5, 6)) == 1
        for a, r in zip(got, ref):
            assert maxerr(a, r) <= 1e-5 * max(1.0, float(r.abs().max())), mode


@pytest.mark.parametrize("N,G,cins,H,W", [(2, 3, [5, 12], 20, 28), (1, 1, [3], 9, 7), (3, 24, [12, 12], 10, 10)])
def test_split_packed_image_is_hi_plus_residual(N, G, cins, H, W):
    """jaf_conv2d_pack_input with d.precision = JAF_PREC_BF16X3: the image holds, per group of 8 channels, a hi plane
    bf16(v) and right behind it a lo plane bf16(v - hi), [N][G][ng8][hi, lo][H*W][8]; hi is bit for bit the bf16 image
    of the same call in bf16 mode, hi + lo reproduces v to 2^-16 relative, channels beyond Cin are zero in both planes."""
    ops = _ops()
    from jafpro_amd.ops import _make_desc
    srcs = [dev(R(880 + i, N, G * c, H, W, lo=-3.0, hi=3.0)) for i, c in enumerate(cins)]
    Cin = sum(cins)
    specs = [(c, G * c, 0, c) for c in cins]
    ng8 = (Cin + 7) // 8
    imgs = {}
    for mode in ("bf16", "bf16x3"):
        prev = ops.set_precision(mode)
        try:
            d = _make_desc(N, G, Cin, 8, H, W, H, W, 3, 3, 1, 1, 1, 1, specs, Cin, 0, G * 8, 0, 0, 0.0)
            imgs[mode] = ops.pack_input(srcs, d).clone()
        finally:
            ops.set_precision(prev)
    torch.cuda.synchronize()
    plain = imgs["bf16"].view(torch.bfloat16).view(N, G, ng8, H * W, 8)
    split = imgs["bf16x3"].view(torch.bfloat16).view(N, G, ng8, 2, H * W, 8)
    assert imgs["bf16x3"].numel() == 2 * imgs["bf16"].numel()
    assert torch.equal(split[:, :, :, 0], plain)
    x = torch.cat([t.view(N, G, c, H * W) for t, c in zip(srcs, cins)], 2)                      # [N, G, Cin, HW]
    xp = torch.zeros(N, G, ng8 * 8, H * W, device="cuda")
    xp[:, :, :Cin] = x
    xp = xp.view(N, G, ng8, 8, H * W).permute(0, 1, 2, 4, 3)
    hi, lo = split[:, :, :, 0].float(), split[:, :, :, 1].float()
    assert torch.equal(hi, xp.to(torch.bfloat16).float())
    assert torch.equal(lo, (xp - hi).to(torch.bfloat16).float())
    assert (hi + lo - xp).abs().max().item() <= 2.0 ** -16 * 3.0
    if Cin % 8:
        assert float(split[:, :, -1, :, :, Cin % 8:].float().abs().max()) == 0.0


def test_kernel_names_come_from_the_launch():
    """jaf_kernel_names / jaf_last_kernel_name: while a profiler is set, every convolution-family launch reports the
    instantiation it picked, in rocprofv3's spelling; the rows of bench.py's roofline are keyed by these strings."""
    import re
    ops = _ops()
    x = dev(R(1, 2, 24, 40, 40))
    w = dev(R(2, 32, 24, 3, 3, lo=-0.1, hi=0.1)).requires_grad_(True)
    seen = {}
    for mode in ("f32", "bf16", "bf16x3"):
        prev = ops.set_precision(mode)
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
        try:
            y = ops.conv2d(x.clone().requires_grad_(True), w, None, stride=1, pad=1, act=1, slope=0.2)
            y.sum().backward()
        finally:
            ops.set_profiler(None)
            ops.set_precision(prev)
        seen[mode] = sorted(prof.summary())
    assert any(re.fullmatch(r"conv_mfma_kernel<\d, \d, \d, false>", k) for k in seen["f32"]), seen["f32"]
    assert any(re.fullmatch(r"conv_dma_kernel<\d, \d, false, (true|false), (true|false)>", k) for k in seen["bf16"]), seen["bf16"]
    assert any(re.fullmatch(r"conv_wgrad_fast_kernel<\d, (true|false), \d, false>", k) for k in seen["bf16"]), seen["bf16"]     # stride-1 3 x 3
    assert any(re.fullmatch(r"conv_dma_split_kernel<\d, \d, false, (true|false)>", k) for k in seen["bf16x3"]), seen["bf16x3"]
    assert any(re.fullmatch(r"conv_wgrad_fast_kernel<\d, (true|false), \d, true>", k) for k in seen["bf16x3"]), seen["bf16x3"]
    assert "conv_pack_dz_kernel" in seen["bf16"] and "conv_pack_input_kernel" in seen["bf16x3"]
    # a stride-2 layer stays on the general weight-gradient kernel
    x = dev(R(3, 1, 24, 40, 40))
    w = dev(R(4, 32, 24, 3, 3, lo=-0.1, hi=0.1)).requires_grad_(True)
    prev = ops.set_precision("bf16")
    prof = ops.KernelProfiler()
    ops.set_profiler(prof)
    try:
        ops.conv2d(x.clone().requires_grad_(True), w, None, stride=2, pad=1, act=0).sum().backward()
    finally:
        ops.set_profiler(None)
        ops.set_precision(prev)
    assert any(re.fullmatch(r"conv_wgrad_dma_kernel<\d, 3, false, (true|false), \d+, false>", k) for k in prof.summary()), sorted(prof.summary())


def test_weight_gradient_partials_are_bit_reproducible():
    """jaf_conv2d_wgrad_packed_ws: with a workspace the pixel splits of a large layer store their blocks of dW to their own copies and
    one pass adds them in a fixed order -- the same bits on every run, and the atomics' result to rounding."""
    ops = _ops()
    x = dev(R(1, 2, 24 * 96, 25, 25))
    w = dev(R(2, 24 * 192, 96, 3, 3, lo=-0.05, hi=0.05))
    proj = dev(R(3, 2, 24 * 192, 25, 25))

    def grad(partials):
        prev_p, prev = ops.set_wgrad_partials(partials), ops.set_precision("bf16")
        try:
            wd = w.clone().requires_grad_(True)
            y = ops.conv2d(x.clone().requires_grad_(True), wd, None, stride=1, pad=1, act=0, groups=24)
            (y * proj).sum().backward()
            torch.cuda.synchronize()
            return wd.grad.clone()
        finally:
            ops.set_precision(prev)
            ops.set_wgrad_partials(prev_p)

    a, b, c = grad(True), grad(True), grad(False)
    import ctypes
    prev = ops.set_precision("bf16")
    d = ops._make_desc(2, 24, 96, 192, 25, 25, 25, 25, 3, 3, 1, 1, 1, 1, [(96, 24 * 96, 0, 96)], 96, 0, 24 * 192, 0, 0, 0.0)
    ops.set_precision(prev)
    assert ops.lib().jaf_conv2d_wgrad_packed_ws_bytes(ctypes.byref(d), 0) > 0      # this layer does take the workspace
    assert torch.equal(a, b)                                        # fixed summation order
    assert maxerr(a, c) <= 1e-4 * max(1.0, c.abs().max().item())    # same sums as the atomics, to fp32 rounding


def test_split_batchnorm_one_launch_is_bit_identical_to_per_chunk_calls():
    """jaf_batchnorm_act_{fwd,bwd}_split: the chunks of a batch (the discriminators' real / generated halves) normalised in ONE launch
    give the bits of the per-chunk launches: outputs, batch and running statistics, input / weight / bias gradients."""
    ops = _ops()
    for (N, C, H, W, parts) in ((16, 64, 64, 64, 2), (6, 24, 9, 7, 3), (8, 256, 4, 4, 2)):
        x0 = dev(R(1, N, C, H, W))
        w0, b0 = dev(R(2, C, lo=0.5, hi=1.5)), dev(R(3, C))
        g = dev(R(4, N, C, H, W))
        res = []
        for one in (True, False):
            prev = ops._SPLIT_BN_ONE_LAUNCH
            ops._SPLIT_BN_ONE_LAUNCH = one
            try:
                x = x0.clone().requires_grad_(True)
                w, b = w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
                rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
                y = ops.batchnorm_act(x, w, b, rm, rv, True, ops.ACT_LRELU, 0.2, batch_parts=parts)
                y.backward(g)
                torch.cuda.synchronize()
                res.append([t.detach().clone() for t in (y, rm, rv, x.grad, w.grad, b.grad)])
            finally:
                ops._SPLIT_BN_ONE_LAUNCH = prev
        for a, r, name in zip(res[0], res[1], ("y", "running_mean", "running_var", "dx", "dweight", "dbias")):
            assert torch.equal(a, r), (N, C, H, W, parts, name)


def test_bce_pair_and_fused_linear_backward_equal_the_separate_launches():
    """ops.bce_pair (two BCE terms of one vector + their sum, one launch each way) against two ops.bce_loss calls on slices, and the
    classifier's fused backward (activation backward + in-place parameter gradients) against act_bwd + linear_bwd + accumulation: same bits."""
    ops = _ops()
    x0 = dev(R(1, 16, 300))
    w0, b0 = dev(R(2, 100, 300, lo=-0.1, hi=0.1)), dev(R(3, 100, lo=-0.1, hi=0.1))
    w1, b1 = dev(R(4, 1, 100, lo=-0.3, hi=0.3)), dev(R(5, 1))
    res = []
    for fused in (True, False):
        prev = ops._LINEAR_FUSED_BWD
        ops._LINEAR_FUSED_BWD = fused
        try:
            ps = [t.clone().requires_grad_(True) for t in (w0, b0, w1, b1)]
            for t in ps:
                t.grad = torch.full_like(t, 0.25)            # the trainers' gradients live in pre-existing buffers
            x = x0.clone().requires_grad_(True)
            h = ops.linear(x, ps[0], ps[1], ops.ACT_LRELU, 0.2)
            p = ops.linear(h, ps[2], ps[3], ops.ACT_SIGMOID, 0.0)
            if fused:
                l1, l2, ls = ops.bce_pair(p, 8, 1.0, 0.0)
                ops.backward_from(ls)
            else:
                l1, l2 = ops.bce_loss(p[:8], 1.0), ops.bce_loss(p[8:], 0.0)
                ls = l1 + l2
                ls.backward()
            torch.cuda.synchronize()
            res.append([t.detach().clone() for t in (l1, l2, ls, x.grad, ps[0].grad, ps[1].grad, ps[2].grad, ps[3].grad)])
        finally:
            ops._LINEAR_FUSED_BWD = prev
    for a, r, name in zip(res[0], res[1], ("real", "fake", "sum", "dx", "dw0", "db0", "dw1", "db1")):
        assert torch.equal(a, r), (name, (a - r).abs().max().item())


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_atlas_slicing_into_the_packed_image_is_bit_identical(mode):
    """ops.atlas_to_parts_packed writes enc1's packed input image itself (train/4...py:269-276 + the packing pass of the first
    part-encoder convolution in one kernel): the accumulate net's output must not change by a bit against
    atlas_to_parts + jaf_conv2d_pack_input, in both packed arithmetics; off the packed path it declines (None)."""
    from jafpro_amd import ops, synth
    from jafpro_amd.networks import Accumulate_LSTM_no_loss
    m = synth.load_synth(Accumulate_LSTM_no_loss(), 21).cuda()
    atlas = torch.from_numpy(synth.uniform(23, "atlas", (1, 2, 3, 800, 1200))).cuda()
    prev = ops.set_precision("f32")
    try:
        assert ops.atlas_to_parts_packed(atlas) is None
        ops.set_precision(mode)
        with torch.no_grad():
            ref = m.forward_grouped(ops.atlas_to_parts(atlas), 2)
            x, img = ops.atlas_to_parts_packed(atlas)
            assert tuple(x.shape) == (2, 72, 200, 200) and img.split == (mode == "bf16x3")
            out = m.forward_grouped(x, 2, x_image=img)
        assert torch.equal(out, ref)
    finally:
        ops.set_precision(prev)


# ------------------------------------------------------------------------------------------------
# bf16 STORAGE of the "bf16" arithmetic mode (ops.bf16_storage_active: BASELINE configs[2] "bf16 storage / fp32 accumulate")
# ------------------------------------------------------------------------------------------------
def _rel(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_bf16_storage_conv_output_is_the_rounded_fp32_output():
    """jaf_packed_io.out_bf16: the same launch with its NCHW result stored in bf16 must hold exactly the RNE rounding of what it
    stores in fp32 (plain epilogue, the statistics epilogue of the CRN LayerNorm, and the accumulate form), and the LayerNorm
    statistics taken in the epilogue must not change (they are summed from the unrounded values)."""
    ops = _ops()
    from jafpro_amd._lib import ACT_NONE
    prev = ops.set_precision("bf16")
    try:
        x, w, b = dev(R(1, 2, 24, 32, 32)), dev(R(2, 40, 24, 3, 3, lo=-0.2, hi=0.2)), dev(R(3, 40))
        with torch.no_grad():
            y32 = ops.conv2d(x, w, b, stride=1, pad=1, act=ACT_NONE)
            y16 = ops.conv2d(x, w, b, stride=1, pad=1, act=ACT_NONE, out_dtype=torch.bfloat16)
            assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.to(torch.bfloat16))
            s32, s16 = ops.LNStats(), ops.LNStats()
            z32 = ops.conv2d(x, w, b, stride=1, pad=1, act=ACT_NONE, ln_stats=s32)
            z16 = ops.conv2d(x, w, b, stride=1, pad=1, act=ACT_NONE, ln_stats=s16, out_dtype=torch.bfloat16)
            assert s32.filled and s16.filled and torch.equal(z16, z32.to(torch.bfloat16))
            a, c = s32.sums.view(2, -1, 2).sum(1), s16.sums.view(2, -1, 2).sum(1)
            assert torch.allclose(a, c, rtol=1e-6, atol=1e-6)
            # 13 x 13 (odd plane: the scalar epilogue) and a strided 2-group layer
            x2, w2 = dev(R(4, 3, 16, 13, 13)), dev(R(5, 24, 8, 3, 3, lo=-0.2, hi=0.2))
            q32 = ops.conv2d(x2, w2, None, stride=1, pad=1, act=ACT_NONE, groups=2)
            q16 = ops.conv2d(x2, w2, None, stride=1, pad=1, act=ACT_NONE, groups=2, out_dtype=torch.bfloat16)
            assert q16.dtype == torch.bfloat16 and torch.equal(q16, q32.to(torch.bfloat16))
            # an activated output stays fp32 (its backward reads y in fp32)
            assert ops.conv2d(x2, w2, None, stride=1, pad=1, act=1, slope=0.2, groups=2, out_dtype=torch.bfloat16).dtype == torch.float32
        off = ops.set_bf16_storage(False)
        try:
            with torch.no_grad():
                assert ops.conv2d(x, w, b, stride=1, pad=1, act=ACT_NONE, out_dtype=torch.bfloat16).dtype == torch.float32
        finally:
            ops.set_bf16_storage(off)
    finally:
        ops.set_precision(prev)
    with torch.no_grad():                                   # the parity-grade modes never store bf16
        assert ops.conv2d(x, w, b, stride=1, pad=1, act=ACT_NONE, out_dtype=torch.bfloat16).dtype == torch.float32


def test_bf16_storage_crn_block_and_convlstm_agree_with_fp32_storage():
    """The CRN conv -> LayerNorm -> LeakyReLU block (src/crn_model.py:90-106) and the ConvLSTM (src/convLSTM.py:41-56) in bf16
    arithmetic with bf16 storage (pre-LayerNorm output and its incoming gradient; cell state, d h and d c of the time loop) against
    the same arithmetic with fp32 storage: outputs within 2e-2 of the output scale, every gradient within 8e-2 relative L2 -- the size
    of one more bf16 rounding per stored tensor (a wrong element type or offset is O(1))."""
    ops = _ops()
    from jafpro_amd import synth
    from jafpro_amd.crn_model import ConvBlock
    prev = ops.set_precision("bf16")
    res = {}
    try:
        for storage in (False, True):
            pst = ops.set_bf16_storage(storage)
            try:
                torch.manual_seed(5)
                blk = synth.load_synth(ConvBlock(2, 9, 32, (3, 3), 1), 77).cuda()
                x = dev(R(6, 2, 9, 32, 32)).requires_grad_(True)
                y = blk(x)
                assert y.dtype == torch.float32
                (y * dev(R(7, *y.shape))).sum().backward()
                g = [x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
                # ConvLSTM: T = 4, 2 groups of 8 hidden channels on 20 x 20 and the odd 13 x 13 plane
                outs = []
                for hw in (20, 13):
                    xs = dev(R(8, 4, 2, 16, hw, hw)).requires_grad_(True)
                    w = dev(R(9, 2 * 32, 16, 3, 3, lo=-0.15, hi=0.15)).requires_grad_(True)
                    b = dev(R(10, 2 * 32)).requires_grad_(True)
                    h, _ = ops.convlstm(xs, w, b, groups=2, return_all=False, return_state=False)
                    (h * dev(R(11, *h.shape))).sum().backward()
                    outs += [h.detach().clone(), xs.grad.clone(), w.grad.clone(), b.grad.clone()]
                res[storage] = (y.detach().clone(), g, outs)
            finally:
                ops.set_bf16_storage(pst)
    finally:
        ops.set_precision(prev)
    (y0, g0, o0), (y1, g1, o1) = res[False], res[True]
    assert (y1 - y0).abs().max().item() <= 2e-2 * max(1.0, y0.abs().max().item())
    # (measured on MI355X: LayerNorm-block gradients up to 4.4e-2 -- a rounded x flips LeakyReLU(0.01) decisions of elements
    # near zero, as the bf16 arithmetic itself does against fp32: tests/test_gpu_step_parity.py BF16_GRAD_BARS 0.12-0.30, unchanged)
    for i, (a, b) in enumerate(zip(g1, g0)):
        print("crn block gradient %d: rel-L2 %.3e" % (i, _rel(a, b)))
        assert _rel(a, b) <= 8e-2, ("crn block gradient", i, _rel(a, b))
    for i, (a, b) in enumerate(zip(o1, o0)):
        print("convlstm result %d: rel-L2 %.3e" % (i, _rel(a, b)))
        assert _rel(a, b) <= 8e-2, ("convlstm", i, _rel(a, b))
    assert any(not torch.equal(a, b) for a, b in zip(o1, o0)), "bf16 storage left every ConvLSTM result bit-identical: it did not run"


@pytest.mark.parametrize("mode", ["bf16", "bf16x3"])
def test_batched_weight_repack_equals_a_fresh_pack(mode):
    """ops.refresh_packed_weights re-makes the cached weight images of a module in ONE launch whose threads take all taps of 8 channels of
    a row (csrc/conv_pack_weights.hip: jafb_repack_rows) and rely on the zero padding the first pack left.  After the weights have
    changed, every re-made image -- forward, data-gradient and both ConvLSTM forms, ragged channel counts, grouped, 1x1 / 3x3 / 5x5 / 7x7 --
    must equal, byte for byte, an image packed from scratch by jaf_conv2d_pack."""
    import ctypes
    ops = _ops()
    from jafpro_amd._lib import lib
    prev = ops.set_precision(mode)
    try:
        ws = []
        cases = [(2, 1, 5, 7, 12, 3), (1, 1, 70, 128, 16, 3), (2, 3, 4, 12, 10, 5), (1, 1, 9, 32, 20, 7), (1, 1, 256, 3, 8, 1), (1, 24, 24, 6, 12, 3),
                 (1, 1, 515, 512, 4, 3)]
        for i, (N, G, Cin, Cout, S, k) in enumerate(cases):
            x = dev(R(20 + i, N, G * Cin, S, S)).requires_grad_(True)
            w = dev(R(40 + i, G * Cout, Cin, k, k, lo=-0.3, hi=0.3)).requires_grad_(True)
            b = dev(R(60 + i, G * Cout)).requires_grad_(True)
            y = ops.conv2d(x, w, b, stride=1, pad=k // 2, act=1, slope=0.2, groups=G)
            y.sum().backward()
            ws.append(w)
        T, N, G, C, S = 2, 1, 2, 12, 10
        xl = dev(R(80, T, N, G * C, S, S)).requires_grad_(True)
        wl = dev(R(81, G * 4 * C, 2 * C, 3, 3, lo=-0.3, hi=0.3)).requires_grad_(True)
        bl = dev(R(82, G * 4 * C)).requires_grad_(True)
        h, _ = ops.convlstm(xl, wl, bl, groups=G)
        h.sum().backward()
        ws.append(wl)
        n_entries = 0
        with torch.no_grad():
            for w in ws:
                w.mul_(1.37).add_(0.011)
        ops.refresh_packed_weights(ws)
        torch.cuda.synchronize()
        L = lib()
        for w in ws:
            for ck in ops._PACK_KEYS_BY_ID.get(id(w), ()):
                e = ops._PACK_CACHE.get(ck)
                if e is None or e.ref() is not w:
                    continue
                fresh = torch.empty_like(e.buf)
                fresh.fill_(float("nan"))
                ops.check(L.jaf_conv2d_pack(ops._s(), ctypes.byref(e.d), ctypes.byref(e.pl), e.mode, ops._p(w), e.rows, ops._p(fresh)), "jaf_conv2d_pack")
                torch.cuda.synchronize()
                assert torch.equal(e.buf.view(torch.int32), fresh.view(torch.int32)), (tuple(w.shape), e.mode)
                n_entries += 1
        assert n_entries >= 2 * len(cases)          # forward + data-gradient image of every convolution (+ the ConvLSTM's)
    finally:
        ops.set_precision(prev)
        ops.invalidate_packed_weights()
