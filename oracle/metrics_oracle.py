"""ORACLE -- TEST INFRASTRUCTURE ONLY.  NumPy / SciPy restatement of the metrics of test/video_evaluation.py:165-214.
The libraries that script calls are absent here (cv2, scikit-image 0.16.2, scikit-video 1.1.11: requirements.txt), so
each function restates the PUBLISHED algorithm of the call it stands for and is pinned by closed-form known answers in
tests/test_oracle_golden.py (identical frames, constant offsets, pure colours); MS-SSIM is "parity unpinned" against
scikit-video's border handling."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import uniform_filter
from scipy.signal import correlate2d


def bgr_to_gray(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) for uint8: 14-bit fixed-point coefficients, round to nearest."""
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + 8192) >> 14).astype(np.uint8)


def compare_ssim(X, Y):
    """skimage.measure.compare_ssim(X, Y) defaults for uint8 2-D input (skimage/metrics/_structural_similarity.py, 0.16.2)."""
    K1, K2, win = 0.01, 0.03, 7
    X, Y = X.astype(np.float64), Y.astype(np.float64)
    NP = win ** 2
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(X, size=win), uniform_filter(Y, size=win)
    uxx, uyy, uxy = uniform_filter(X * X, size=win), uniform_filter(Y * Y, size=win), uniform_filter(X * Y, size=win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    R = 255
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win - 1) // 2
    return S[pad:-pad, pad:-pad].mean()


def psnr(ref, dist):
    mse = np.mean((ref.astype(np.float64) - dist.astype(np.float64)) ** 2)
    return 10 * np.log10(255.0 ** 2 / mse)


def msssim(ref, dist):
    """Wang, Simoncelli, Bovik 2003: five scales, 11x11 Gaussian sigma 1.5 (valid windows), 2x2 box down-sampling."""
    weights = [0.0448, 0.2856, 0.3001, 0.2363, 0.1333]
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    w = np.outer(g, g)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    x, y = ref.astype(np.float32).astype(np.float64), dist.astype(np.float32).astype(np.float64)
    out = 1.0
    for i, wt in enumerate(weights):
        f = lambda a: correlate2d(a, w, mode="valid")
        ux, uy = f(x), f(y)
        vx, vy, vxy = f(x * x) - ux * ux, f(y * y) - uy * uy, f(x * y) - ux * uy
        cs = (2 * vxy + C2) / (vx + vy + C2)
        s = (2 * ux * uy + C1) / (ux ** 2 + uy ** 2 + C1) * cs
        out *= (s.mean() if i == 4 else cs.mean()) ** wt
        if i < 4:
            H, W = x.shape[0] // 2 * 2, x.shape[1] // 2 * 2
            pool = lambda a: a[:H, :W].reshape(H // 2, 2, W // 2, 2).astype(np.float32).mean((1, 3), dtype=np.float32).astype(np.float64)
            x, y = pool(x), pool(y)
    return out


def l1_normalised(pred_bgr, gt_bgr):
    p = ((pred_bgr[..., ::-1] / 255. - 0.5) * 2).astype(np.float32)
    g = ((gt_bgr[..., ::-1] / 255. - 0.5) * 2).astype(np.float32)
    return float(np.abs(p - g).mean(dtype=np.float64))
