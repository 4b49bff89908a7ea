"""device evaluation metrics vs oracle/metrics_oracle.py (test/video_evaluation.py:165-214)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _video(seed, F=4, S=256):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:S, 0:S]
    gt = np.zeros((F, S, S, 3), np.uint8)
    for f in range(F):
        base = 127 + 100 * np.sin(xx / 17.0 + f) * np.cos(yy / 23.0 - f)
        for c in range(3):
            gt[f, ..., c] = np.clip(base + 20 * c + r.normal(0, 6, (S, S)), 0, 255)
    pred = np.clip(gt.astype(np.int64) + r.integers(-12, 13, gt.shape), 0, 255).astype(np.uint8)
    pred[:, 100:140, 60:120] = 255 - pred[:, 100:140, 60:120]         # a structurally wrong region
    return pred, gt


def test_metrics_match_the_oracle():
    from jafpro_amd import evaluation as E
    from oracle import metrics_oracle as MO
    pred, gt = _video(0)
    p, g = torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda()
    pg, gg = E.bgr_to_gray(p), E.bgr_to_gray(g)
    assert np.array_equal(pg.cpu().numpy(), MO.bgr_to_gray(pred)) and np.array_equal(gg.cpu().numpy(), MO.bgr_to_gray(gt))
    s, ms, ps, l1 = E.ssim(pg, gg).cpu().numpy(), E.msssim(gg, pg).cpu().numpy(), E.psnr(gg, pg).cpu().numpy(), E.l1_normalised(p, g).cpu().numpy()
    for f in range(pred.shape[0]):
        a, b = MO.bgr_to_gray(pred[f]), MO.bgr_to_gray(gt[f])
        assert abs(s[f] - MO.compare_ssim(a, b)) <= 1e-9, (f, s[f], MO.compare_ssim(a, b))
        assert abs(ms[f] - MO.msssim(b, a)) <= 1e-6, (f, ms[f], MO.msssim(b, a))
        assert abs(ps[f] - MO.psnr(b, a)) <= 1e-9
        assert abs(l1[f] - MO.l1_normalised(pred[f], gt[f])) <= 1e-6
    assert 0.2 < s.mean() < 0.98 and 15 < ps.mean() < 40
    one = E.ssim(gg, gg).cpu().numpy()
    assert np.abs(one - 1.0).max() <= 1e-12 and np.abs(E.msssim(gg, gg).cpu().numpy() - 1.0).max() <= 1e-9


def test_video_evaluator_end_to_end():
    """All five figures of one video, the VGG term against the oracle's VGG features on the same synthetic weights with the
    evaluation script's RGB preprocessing (video_evaluation.py:19-25,191)."""
    import torch.nn.functional as F
    from jafpro_amd import evaluation as E, synth
    from oracle import metrics_oracle as MO
    from oracle import torch_oracle as O
    pred, gt = _video(1, F=2, S=192)
    ev = E.VideoEvaluator()
    assert not ev.flow_weights_loaded          # "flow" is left out until FlowNet2-SD weights arrive (random weights mean nothing)
    # a partial, non-strict load that carries only SOME flow keys must not count as loaded
    part = {k: v for i, (k, v) in enumerate(ev.state_dict().items()) if not (k.startswith("flow_criterion.") and i % 2)}
    ev.load_state_dict(part, strict=False)
    assert not ev.flow_weights_loaded
    o0 = ev.cuda()(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
    assert "flow" not in o0 and all(np.isfinite(v) for v in o0.values())
    assert set(E.VideoEvaluator.aggregate([o0, dict(o0, flow=2.0), dict(o0, flow=4.0)])) == set(o0) | {"flow"}
    assert E.VideoEvaluator.aggregate([o0, dict(o0, flow=2.0), dict(o0, flow=4.0)])["flow"] == 3.0
    ev = ev.cpu()
    synth.load_synth(ev, 808)
    assert ev.flow_weights_loaded
    assert E.VideoEvaluator(with_flow=False).flow_criterion is None
    sd = {k: v.detach().clone() for k, v in ev.state_dict().items()}
    ev = ev.cuda()
    out = ev(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
    ref_vgg = 0.0
    mean = torch.tensor([123.68, 116.779, 103.939]).view(1, 3, 1, 1)
    for f in range(2):
        p = torch.from_numpy(((pred[f][..., ::-1] / 255. - 0.5) * 2).astype(np.float32).copy()).permute(2, 0, 1)[None]
        g = torch.from_numpy(((gt[f][..., ::-1] / 255. - 0.5) * 2).astype(np.float32).copy()).permute(2, 0, 1)[None]
        fp = O.vgg_features(sd, 255.0 * (p + 1.0) / 2.0 - mean, pre="perceptual_criterion.vgg.vgg_model.")
        fg = O.vgg_features(sd, 255.0 * (g + 1.0) / 2.0 - mean, pre="perceptual_criterion.vgg.vgg_model.")
        ref_vgg += sum(w * float(F.l1_loss(a, b)) for w, a, b in zip(O.VGG_WEIGHTS, fp, fg))
    assert abs(out["vgg"] - ref_vgg / 2) <= 1e-3 * ref_vgg / 2
    assert abs(out["ssim"] - np.mean([MO.compare_ssim(MO.bgr_to_gray(pred[f]), MO.bgr_to_gray(gt[f])) for f in range(2)])) <= 1e-9
    assert set(out) == {"ssim", "msssim", "psnr", "l1", "vgg", "flow"} and all(np.isfinite(v) for v in out.values())
    # the temporal term: FlowNetSD on the two consecutive-frame pairs (:197-206), oracle = functional restatement pinned
    # bit-exactly on the imported reference module (oracle/make_golden.py g_flownet)
    fsd = {k[len("flow_criterion."):]: v for k, v in sd.items() if k.startswith("flow_criterion.")}
    rgb = lambda a: torch.from_numpy(((a[..., ::-1] / 255. - 0.5) * 2).astype(np.float32).copy()).permute(0, 3, 1, 2)
    pr, gr = rgb(pred), rgb(gt)
    with torch.no_grad():
        a = O.flownet_sd_forward(fsd, torch.cat([pr[:-1], pr[1:]], 1) / 2.0 + 0.5)[0]
        b = O.flownet_sd_forward(fsd, torch.cat([gr[:-1], gr[1:]], 1) / 2.0 + 0.5)[0]
    ref_flow = float(F.l1_loss(a, b)) / 2
    assert ref_flow > 0 and abs(out["flow"] - ref_flow) <= 1e-3 * ref_flow, (out["flow"], ref_flow)


def test_flownet_sd_golden(golden_dir):
    """FlowNetSD (src/flownet2_pytorch/networks/FlowNetSD.py:11-106, batchNorm=False) vs the golden recorded from the
    imported reference module on the same synthetic weights: all five flows in train mode, flow2 alone in eval mode;
    conv_transpose2d vs torch for the shapes the network uses; state_dict keys / shapes = the reference's."""
    import json
    import os
    import torch.nn.functional as F
    from jafpro_amd import ops, synth
    from jafpro_amd.flownet_sd import FlowNetSD
    st = dict(np.load(os.path.join(golden_dir, "flownet_sd.npz")))
    m = FlowNetSD(args=[], batchNorm=False)
    schema = json.load(open(os.path.join(golden_dir, "state_dict_schema.json")))["FlowNetSD"]
    assert [[k, list(v.shape)] for k, v in m.state_dict().items()] == schema
    synth.load_synth(m, 811)
    m = m.cuda()
    pairs = torch.from_numpy(st["pairs"]).cuda()
    flows = m(pairs)
    assert len(flows) == 5
    for i, f in enumerate(flows):
        ref = torch.from_numpy(st["flow%d" % (i + 2)])
        err = (f.cpu() - ref).abs().max().item()
        print("flow%d max|diff| %.3e (ref max %.3f)" % (i + 2, err, ref.abs().max().item()))
        assert err <= 1e-3 * max(1.0, ref.abs().max().item()), (i, err)
    m.eval()
    assert len(m(pairs)) == 1 and torch.equal(m(pairs)[0], flows[0])
    l1 = float(ops.l1_loss(flows[0][:1].contiguous(), flows[0][1:].contiguous()))
    assert abs(l1 - float(st["flow_l1"])) <= 1e-3 * float(st["flow_l1"])
    # the transposed convolution alone, odd channel counts and sizes
    g = torch.Generator().manual_seed(5)
    for cin, cout, H, W in ((2, 2, 5, 7), (19, 11, 6, 4), (130, 64, 8, 12)):
        x, w, b = torch.rand(2, cin, H, W, generator=g) - 0.5, torch.rand(cin, cout, 4, 4, generator=g) - 0.5, torch.rand(cout, generator=g)
        ref = F.leaky_relu(F.conv_transpose2d(x, w, b, stride=2, padding=1), 0.1)
        with torch.no_grad():
            out = ops.conv_transpose2d(x.cuda(), w.cuda(), b.cuda(), 2, 1, ops.ACT_LRELU, 0.1)
        assert out.shape == ref.shape and (out.cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    with pytest.raises(RuntimeError):
        ops.conv_transpose2d(x.cuda().requires_grad_(True), w.cuda(), b.cuda())
