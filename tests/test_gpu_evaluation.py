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
    synth.load_synth(ev, 808)
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
    assert set(out) == {"ssim", "msssim", "psnr", "l1", "vgg"} and all(np.isfinite(v) for v in out.values())
