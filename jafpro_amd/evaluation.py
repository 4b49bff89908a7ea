"""Evaluation path on the device: the per-video metrics of test/video_evaluation.py:150-212 -- SSIM,
MS-SSIM and PSNR on the grayscale frames, L1 and the VGG perceptual distance on the normalised RGB frames -- computed
for a whole video at once from the decoded uint8 frames.

Third-party arithmetic (absent from /root/reference and from this image; restated from the published algorithms, pinned
versions from requirements.txt):
  * OpenCV `cvtColor(COLOR_BGR2GRAY)`: 14-bit fixed point 0.114 B + 0.587 G + 0.299 R with rounding;
  * scikit-image 0.16.2 `compare_ssim(x, y)` on uint8: 7x7 uniform window, sample covariance (NP/(NP-1)), K1 0.01, K2 0.03,
    data_range 255, mean over the map cropped by 3 pixels;
  * scikit-video 1.1.11 `psnr`: 10 log10(255^2 / mse) per frame; `msssim`: Wang-Simoncelli-Bovik multi-scale SSIM, five
    scales, 11x11 Gaussian (sigma 1.5) windows, exponents 0.0448 0.2856 0.3001 0.2363 0.1333, 2x2 box down-sampling
    -- PARITY UNPINNED against scikit-video's own border handling (its source is not available offline).
The FlowNetSD temporal term (:66-67,197-206) runs on jafpro_amd.flownet_sd.FlowNetSD (pure convolutions: none of FlowNet2's
CUDA extensions are involved); its FlowNet2-SD checkpoint is an external download, so weights come from load_state_dict.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional

import numpy as np
import torch

from . import ops
from ._lib import check, lib
from .flownet_sd import FlowNetSD, flownet_preprocess
from .networks import VGGLoss_CRN

MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def bgr_to_gray(frames_u8: torch.Tensor) -> torch.Tensor:
    """uint8 [..., H, W, 3] (BGR, as cv2.imread returns) -> uint8 [..., H, W]."""
    ops._chk(frames_u8, "frames", torch.uint8)
    out = torch.empty(frames_u8.shape[:-1], device=frames_u8.device, dtype=torch.uint8)
    check(lib().jaf_bgr_to_gray_u8(ops._s(), ops._p(frames_u8), ops._p(out), out.numel()), "jaf_bgr_to_gray_u8")
    return out


def _window_sums(x: torch.Tensor, y: torch.Tensor, w: np.ndarray, cov_norm: float, C1: float, C2: float) -> torch.Tensor:
    N, H, W = x.shape
    win = int(round(len(w) ** 0.5))
    wd = torch.from_numpy(np.ascontiguousarray(w, np.float64)).to(x.device)
    sums = torch.zeros((N, 2), device=x.device, dtype=torch.float64)
    check(lib().jaf_ssim_window_sums(ops._s(), ops._p(x), ops._p(y), ctypes.c_void_p(wd.data_ptr()), ctypes.c_void_p(sums.data_ptr()),
                                     N, H, W, win, cov_norm, C1, C2), "jaf_ssim_window_sums")
    return sums / float((H - win + 1) * (W - win + 1))


def ssim(pred_gray: torch.Tensor, gt_gray: torch.Tensor) -> torch.Tensor:
    """skimage.measure.compare_ssim of uint8 gray frames [N,H,W] -> fp64 [N] (video_evaluation.py:184)."""
    x, y = pred_gray.float().contiguous(), gt_gray.float().contiguous()
    w = np.full(49, 1.0 / 49.0)
    return _window_sums(x, y, w, 49.0 / 48.0, (0.01 * 255) ** 2, (0.03 * 255) ** 2)[:, 0]


def gaussian_window(size: int = 11, sigma: float = 1.5) -> np.ndarray:
    g = np.exp(-((np.arange(size) - (size - 1) / 2.0) ** 2) / (2.0 * sigma * sigma))
    g /= g.sum()
    return np.outer(g, g).reshape(-1)


def msssim(gt_gray: torch.Tensor, pred_gray: torch.Tensor) -> torch.Tensor:
    """Multi-scale SSIM of uint8 gray frames [N,H,W] -> fp64 [N] (video_evaluation.py:209)."""
    x, y = gt_gray.float().contiguous(), pred_gray.float().contiguous()
    w = gaussian_window()
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    out = None
    for i, wt in enumerate(MSSSIM_WEIGHTS):
        m = _window_sums(x, y, w, 1.0, C1, C2)
        term = (m[:, 0] if i == len(MSSSIM_WEIGHTS) - 1 else m[:, 1]) ** wt
        out = term if out is None else out * term
        if i < len(MSSSIM_WEIGHTS) - 1:
            if x.shape[1] % 2 or x.shape[2] % 2:
                x, y = x[:, :x.shape[1] // 2 * 2, :x.shape[2] // 2 * 2].contiguous(), y[:, :y.shape[1] // 2 * 2, :y.shape[2] // 2 * 2].contiguous()
            x = ops.avg_pool(x.unsqueeze(1), 2, 2, 0)[:, 0].contiguous()
            y = ops.avg_pool(y.unsqueeze(1), 2, 2, 0)[:, 0].contiguous()
    return out


def _error_sums(a_u8: torch.Tensor, b_u8: torch.Tensor) -> torch.Tensor:
    N = a_u8.shape[0]
    P = a_u8.numel() // N
    sums = torch.zeros((N, 2), device=a_u8.device, dtype=torch.float64)
    check(lib().jaf_frame_error_sums_u8(ops._s(), ops._p(a_u8), ops._p(b_u8), ctypes.c_void_p(sums.data_ptr()), N, P),
          "jaf_frame_error_sums_u8")
    return sums


def psnr(gt_gray: torch.Tensor, pred_gray: torch.Tensor) -> torch.Tensor:
    """skvideo.measure.psnr of uint8 gray frames [N,H,W] -> fp64 [N] (video_evaluation.py:213)."""
    s = _error_sums(gt_gray.contiguous(), pred_gray.contiguous())
    mse = s[:, 0] / float(gt_gray[0].numel())
    return 10.0 * torch.log10(255.0 ** 2 / mse)


def l1_normalised(pred_u8: torch.Tensor, gt_u8: torch.Tensor) -> torch.Tensor:
    """nn.L1Loss on the frames normalised to (-1, 1) (video_evaluation.py:174-175,188) -> fp64 [N]: the map is affine, so
    the mean absolute difference is 2/255 of the uint8 one."""
    s = _error_sums(pred_u8.contiguous(), gt_u8.contiguous())
    return s[:, 1] / float(pred_u8[0].numel()) * (2.0 / 255.0)


class VideoEvaluator(torch.nn.Module):
    """perceptual_criterion = VGGLoss_CRN(weights=[1/2.6, 1/4.8, 1/3.7, 1/5.6, 10/1.5]) (video_evaluation.py:66) plus the
    closed-form metrics.  VGG weights come from load_state_dict (the ImageNet weights are not available offline).
    `with_flow=True` adds the script's FlowNetSD temporal term (:66-67): its weights are the external FlowNet2-SD checkpoint --
    until they have been loaded (`load_flow_weights`, or a load_state_dict that carries EVERY `flow_criterion.*` key) the evaluator
    leaves "flow" out of its result instead of a number computed from the random initialisation."""

    def __init__(self, with_flow: bool = True):
        super().__init__()
        self.perceptual_criterion = VGGLoss_CRN(weights=[1 / 2.6, 1 / 4.8, 1 / 3.7, 1 / 5.6, 10 / 1.5])
        # left in train mode like the script: `[0]` = flow2
        self.flow_criterion = FlowNetSD(args=[], batchNorm=False) if with_flow else None
        self.flow_weights_loaded = False

    def load_flow_weights(self, state_dict, strict: bool = True):
        """FlowNet2-SD weights (the reference loads `FlowNet2-SD_checkpoint.pth.tar`['state_dict'], video_evaluation.py:67)."""
        if self.flow_criterion is None:
            raise RuntimeError("VideoEvaluator(with_flow=False) has no flow criterion")
        r = self.flow_criterion.load_state_dict(state_dict, strict=strict)
        self.flow_weights_loaded = not r.missing_keys
        return r

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        """The flow term counts as loaded only when EVERY `flow_criterion.*` key was present (a partial / non-strict load that
        carries a few of them leaves the rest at their random initialisation: ADVICE r4)."""
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        if self.flow_criterion is not None:
            want = ["flow_criterion." + k for k in self.flow_criterion.state_dict()]
            self.flow_weights_loaded = all(k in state_dict for k in want) and not any(k.startswith("flow_criterion.") for k in r.missing_keys)
        return r

    @torch.no_grad()
    def flow_error(self, pred_rgb: torch.Tensor, gt_rgb: torch.Tensor) -> float:
        """Sum over consecutive frame pairs of L1(FlowNetSD(pred pair)[0], FlowNetSD(gt pair)[0]) (:197-206); frames [F,3,H,W]
        RGB in (-1, 1), H and W multiples of 64.  The pairs of one video go through the network as one batch."""
        if pred_rgb.shape[0] < 2:
            return 0.0
        if pred_rgb.shape[-2] % 64 or pred_rgb.shape[-1] % 64:
            raise ValueError("flow_error: FlowNetSD needs frame sizes that are multiples of 64, got %dx%d" % tuple(pred_rgb.shape[-2:]))
        pp = flownet_preprocess(torch.cat([pred_rgb[:-1], pred_rgb[1:]], 1))
        gp = flownet_preprocess(torch.cat([gt_rgb[:-1], gt_rgb[1:]], 1))
        fp, fg = self.flow_criterion(pp)[0], self.flow_criterion(gp)[0]
        per = fp[0].numel()
        tot = 0.0
        for i in range(fp.shape[0]):                                   # nn.L1Loss per pair (a mean), summed over the video
            tot += float(ops.l1_loss(fp[i:i + 1].contiguous(), fg[i:i + 1].contiguous()))
        return tot

    @staticmethod
    def vgg_preprocess_rgb(x: torch.Tensor) -> torch.Tensor:
        """The evaluation script's own vgg_preprocess (:19-25): RGB input, means subtracted from channels 2, 1, 0."""
        x = 255.0 * (x + 1.0) / 2.0
        mean = torch.tensor([123.68, 116.779, 103.939], device=x.device, dtype=x.dtype).view(1, 3, 1, 1)
        return (x - mean).contiguous()

    @torch.no_grad()
    def forward(self, pred_bgr_u8: torch.Tensor, gt_bgr_u8: torch.Tensor) -> Dict[str, float]:
        """uint8 [F,H,W,3] BGR frames of one video -> the per-video sums the script accumulates (:184-214), divided by F
        where it reports means."""
        from .data import normalize_images
        F = pred_bgr_u8.shape[0]
        pg, gg = bgr_to_gray(pred_bgr_u8), bgr_to_gray(gt_bgr_u8)
        out = {"ssim": float(ssim(pg, gg).sum()) / F, "msssim": float(msssim(gg, pg).sum()) / F, "psnr": float(psnr(gg, pg).sum()) / F,
               "l1": float(l1_normalised(pred_bgr_u8, gt_bgr_u8).sum()) / F}
        # BGR -> RGB (:174-175) is a channel flip of the normalised tensor
        p = normalize_images(pred_bgr_u8).flip(1).contiguous()
        g = normalize_images(gt_bgr_u8).flip(1).contiguous()
        vgg = 0.0
        for i in range(F):                                         # the script scores frame by frame (:191)
            vgg += float(self.perceptual_criterion(self.vgg_preprocess_rgb(p[i:i + 1]), self.vgg_preprocess_rgb(g[i:i + 1])))
        out["vgg"] = vgg / F
        # the script divides the F-1 terms by F (:220).  The key is OMITTED while the FlowNet2-SD weights have not been loaded (it
        # was NaN in round 4: a caller that averages over videos or over keys then propagated the NaN; `aggregate` below skips
        # videos without the key)
        if self.flow_criterion is not None and self.flow_weights_loaded:
            out["flow"] = self.flow_error(p, g) / F
        return out

    @staticmethod
    def aggregate(per_video) -> Dict[str, float]:
        """Mean of every metric over the videos that report it (video_evaluation.py:216-222 prints these means)."""
        keys = sorted({k for d in per_video for k in d})
        return {k: float(sum(d[k] for d in per_video if k in d) / max(1, sum(1 for d in per_video if k in d))) for k in keys}
