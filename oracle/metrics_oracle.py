"""TEST INFRASTRUCTURE ONLY - independent float64 numpy/scipy restatement of SSIM (Wang et al. 2004) and
HaarPSI (Reisenhofer et al. 2018) with the conventions of piq==0.8.0 (the reference's dependency,
src/utils/evaluate.py:16,73-76; absent from /root/reference and from this image => parity with piq is
UNPINNED).  Written against the published algorithms, loop/convolve2d-based, sharing no code with
miccai24_immoco_amd/utils/evaluate.py; only tests/ may import it."""
import numpy as np
from scipy.signal import correlate2d


def ssim_np(x, y, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """x, y: 2-D float arrays in [0, 1]; no down-sampling below 384 px (factor round(min/256) == 1)."""
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    f = max(1, round(min(x.shape) / 256))
    if f > 1:
        h, w = (x.shape[0] // f) * f, (x.shape[1] // f) * f
        x = x[:h, :w].reshape(h // f, f, w // f, f).mean((1, 3))
        y = y[:h, :w].reshape(h // f, f, w // f, f).mean((1, 3))
    r = np.arange(kernel_size) - (kernel_size - 1) / 2
    g = np.exp(-(r[:, None] ** 2 + r[None, :] ** 2) / (2 * sigma ** 2))
    g /= g.sum()
    m = lambda a: correlate2d(a, g, mode="valid")
    mx, my = m(x), m(y)
    vx, vy, vxy = m(x * x) - mx ** 2, m(y * y) - my ** 2, m(x * y) - mx * my
    c1, c2 = k1 ** 2, k2 ** 2
    return float(np.mean((2 * mx * my + c1) * (2 * vxy + c2) / ((mx ** 2 + my ** 2 + c1) * (vx + vy + c2))))


def _haar_same(img, j, transpose):
    """MATLAB conv2(img, filt, 'same')-sized Haar response at scale j (value 2^-j, sign irrelevant: abs)."""
    k = 2 ** j
    filt = np.full((k, k), 2.0 ** -j)
    filt[k // 2:, :] *= -1
    if transpose:
        filt = filt.T
    H, W = img.shape
    pad = np.zeros((H + k - 1, W + k - 1))
    pad[k // 2 - 1:k // 2 - 1 + H, k // 2 - 1:k // 2 - 1 + W] = img
    out = np.zeros((H, W))
    for dy in range(k):
        for dx in range(k):
            out += filt[dy, dx] * pad[dy:dy + H, dx:dx + W]
    return np.abs(out)


def haarpsi_np(x, y, c=30.0, alpha=4.2):
    """x, y: 2-D float arrays in [0, 1]; 3 scales, 2x down-sampling as in the paper."""
    def prep(a):
        a = np.asarray(a, np.float64) * 255.0
        odd = max(a.shape[0] % 2, a.shape[1] % 2)
        a = np.pad(a, ((0, odd), (0, odd)))
        h, w = a.shape[0] // 2 * 2, a.shape[1] // 2 * 2
        return a[:h, :w].reshape(h // 2, 2, w // 2, 2).mean((1, 3))
    x, y = prep(x), prep(y)
    num = den = 0.0
    for o in (False, True):
        ax = [_haar_same(x, j, o) for j in (1, 2, 3)]
        ay = [_haar_same(y, j, o) for j in (1, 2, 3)]
        sim = sum((2 * a * b + c) / (a * a + b * b + c) for a, b in zip(ax[:2], ay[:2])) / 2
        wgt = np.maximum(ax[2], ay[2])
        num += np.sum(wgt / (1 + np.exp(-alpha * sim)))
        den += np.sum(wgt)
    eps = np.finfo(np.float32).eps
    s = (num + eps) / (den + eps)
    return float((np.log(s / (1 - s)) / alpha) ** 2)
