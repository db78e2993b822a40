"""The reference's metric path (src/utils/evaluate.py:19-80 and the centre crop of
src/test/test_immoco.py:74-85), evaluated where the tensors live so that a batch of corrected slices never
leaves the device.  Plain tensor arithmetic, outside the hot path.

PSNR/RMSE/normalize are pinned by reference golden vectors.  SSIM and HaarPSI come from `piq`
(piq==0.8.0 in the reference's requirements.txt, absent here): they restate the published algorithms
(Wang et al. 2004; Reisenhofer et al. 2018) with piq's conventions - parity with piq itself is UNPINNED;
tests compare against an independent float64 scipy restatement in oracle/metrics_oracle.py and known answers.
"""

import torch
import torch.nn.functional as F


def normalize(x: torch.Tensor) -> torch.Tensor:
    """Min-max to [0,1]; batch-wise when the batch dimension is > 1 (evaluate.py:19-29)."""
    if x.shape[0] > 1:
        flat = x.reshape(x.shape[0], -1)
        mx, mn = flat.max(1).values, flat.min(1).values
        return (x - mn.view(-1, 1, 1, 1)) / ((mx - mn).view(-1, 1, 1, 1) + 1e-24)
    return (x - x.min()) / (x.max() - x.min() + 1e-24)


def rmse(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return torch.sqrt(torch.mean((x - y) ** 2))


def my_psnr(img1, img2, data_range=None, reduction="mean"):
    mse = torch.mean((img1 - img2) ** 2, dim=(1, 2, 3))
    peak = img2.reshape(img2.shape[0], -1).max(1).values if data_range is None else data_range
    val = 20 * torch.log10(peak / torch.sqrt(mse))
    return val if reduction == "none" else val.mean()


def crop_psnr(pred_abs: torch.Tensor, gt_abs: torch.Tensor) -> float:
    """PSNR as test_immoco.py:74-85 computes it: centre-half crop, min-max normalise, data_range 1."""
    H, W = gt_abs.shape[-2:]
    c0, c1 = int(H / 4), int(W / 4)
    p = pred_abs[c0:-c0, c1:-c1][None, None]
    g = gt_abs[c0:-c0, c1:-c1][None, None]
    return float(my_psnr(normalize(p), normalize(g), data_range=1.0))


def _check_pair(x: torch.Tensor, y: torch.Tensor, data_range: float):
    if x.dim() != 4 or y.dim() != 4 or x.shape != y.shape:
        raise ValueError(f"expected two (N, C, H, W) tensors of one shape, got {tuple(x.shape)} / {tuple(y.shape)}")
    for t in (x, y):
        if float(t.min()) < 0 or float(t.max()) > data_range:
            raise ValueError(f"values outside [0, {data_range}]")


def _reduce(v: torch.Tensor, reduction: str) -> torch.Tensor:
    if reduction == "none":
        return v
    if reduction == "mean":
        return v.mean(0)
    if reduction == "sum":
        return v.sum(0)
    raise ValueError(f"unknown reduction {reduction!r}")


def ssim(x, y, kernel_size: int = 11, kernel_sigma: float = 1.5, data_range: float = 1.0, reduction: str = "mean",
         downsample: bool = True, k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
    """Structural similarity with piq's conventions (as called at evaluate.py:73-75): Gaussian window
    ``kernel_size`` x ``kernel_size`` (sigma 1.5) applied without padding, inputs average-pooled by
    ``round(min(H, W) / 256)`` first when that factor exceeds 1, mean of the SSIM map per channel, then over
    channels, then ``reduction`` over the batch."""
    if kernel_size % 2 != 1:
        raise ValueError(f"kernel size must be odd, got {kernel_size}")
    _check_pair(x, y, data_range)
    x, y = x / float(data_range), y / float(data_range)
    f = max(1, round(min(x.shape[-2:]) / 256))
    if f > 1 and downsample:
        x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    if min(x.shape[-2:]) < kernel_size:
        raise ValueError(f"kernel size {kernel_size} exceeds the image size {tuple(x.shape[-2:])}")
    t = torch.arange(kernel_size, dtype=x.dtype, device=x.device) - (kernel_size - 1) / 2.0
    g = torch.exp(-(t[:, None] ** 2 + t[None, :] ** 2) / (2.0 * kernel_sigma ** 2))
    C = x.shape[1]
    win = (g / g.sum()).expand(C, 1, kernel_size, kernel_size)

    def blur(t):
        return F.conv2d(t, win, groups=C)

    mx, my = blur(x), blur(y)
    sxx, syy, sxy = blur(x * x) - mx * mx, blur(y * y) - my * my, blur(x * y) - mx * my
    c1, c2 = k1 ** 2, k2 ** 2
    smap = (2.0 * mx * my + c1) / (mx * mx + my * my + c1) * ((2.0 * sxy + c2) / (sxx + syy + c2))
    return _reduce(smap.mean(dim=(-1, -2)).mean(1), reduction)


def haarpsi(x, y, reduction: str = "mean", data_range: float = 1.0, scales: int = 3, subsample: bool = True,
            c: float = 30.0, alpha: float = 4.2) -> torch.Tensor:
    """Haar wavelet-based perceptual similarity index (grey-scale path, as called at evaluate.py:76):
    images scaled to [0, 255], 2x2 average-pooled, Haar responses (horizontal/vertical, filter value
    2^-j, 'same'-size output with the extra zero row/column at the bottom/right) at ``scales`` dyadic scales;
    local similarity (2ab + c)/(a^2 + b^2 + c) of the two finest scales averaged per orientation, passed
    through sigmoid(alpha .), weighted by the coarsest-scale response magnitude; inverse-logit squared."""
    _check_pair(x, y, data_range)
    if x.shape[1] != 1:
        raise ValueError("only single-channel images are supported (the reference's use)")
    if scales < 3:
        raise ValueError("scales must be >= 3 (two similarity scales + one weight scale)")
    if min(x.shape[-2:]) < 2 ** (scales + 1):
        raise ValueError(f"images must be at least {2 ** (scales + 1)} pixels wide for {scales} scales")
    x, y = x / float(data_range) * 255.0, y / float(data_range) * 255.0
    if subsample:
        odd = max(x.shape[2] % 2, x.shape[3] % 2)
        x, y = (F.avg_pool2d(F.pad(t, [0, odd, 0, odd]), 2) for t in (x, y))

    def responses(img):
        out = []
        for j in range(1, scales + 1):
            k = 2 ** j
            h = torch.full((k, k), 1.0 / k, dtype=img.dtype, device=img.device)
            h[k // 2:, :] *= -1.0
            w = torch.stack([h, h.t()])[:, None]
            out.append(F.conv2d(F.pad(img, [k // 2 - 1, k // 2, k // 2 - 1, k // 2]), w))
        return torch.cat(out, dim=1).abs()          # (N, 2*scales, H, W): [scale][orientation]

    cx, cy = responses(x), responses(y)
    weights = torch.maximum(cx[:, 4:6], cy[:, 4:6])
    sims = []
    for o in range(2):
        a, b = cx[:, (o, o + 2)], cy[:, (o, o + 2)]
        sims.append(((2.0 * a * b + c) / (a * a + b * b + c)).sum(1, keepdim=True) / 2.0)
    sim = torch.cat(sims, dim=1)
    eps = torch.finfo(sim.dtype).eps
    score = ((torch.sigmoid(alpha * sim) * weights).sum(dim=(1, 2, 3)) + eps) / (weights.sum(dim=(1, 2, 3)) + eps)
    return _reduce((torch.log(score / (1.0 - score)) / alpha) ** 2, reduction)


def calmetric2D(pred_recon: torch.Tensor, gt_recon: torch.Tensor):
    """(PSNR, SSIM, HaarPSI, RMSE) of min-max normalised ``(B, C, H, W)`` tensors (evaluate.py:57-80).
    Images narrower than the 11-pixel SSIM window are refused (the reference's fallback branch there
    reads an unassigned variable and cannot run)."""
    if pred_recon.dim() != 4 or gt_recon.dim() != 4:
        raise ValueError("Input tensors must be 4D")
    pred, gt = normalize(pred_recon), normalize(gt_recon)
    if min(pred.shape[-2:]) < 11:
        raise ValueError("calmetric2D needs images of at least 11 x 11 pixels")
    return (my_psnr(pred, gt, data_range=1.0, reduction="mean"),
            ssim(pred, gt, data_range=1.0, kernel_size=11, reduction="mean"),
            haarpsi(pred, gt, scales=3, reduction="mean"),
            rmse(pred, gt))


def calmetric3D(pred_recon: torch.Tensor, gt_recon: torch.Tensor):
    """Per-slice calmetric2D averaged over the batch (evaluate.py:83-97); results stay on the device."""
    rows = [torch.stack([torch.as_tensor(v, device=pred_recon.device, dtype=torch.float32)
                         for v in calmetric2D(pred_recon[i:i + 1], gt_recon[i:i + 1])])
            for i in range(pred_recon.shape[0])]
    return tuple(torch.stack(rows).mean(0))


def slice_metrics(image: torch.Tensor, image_gt: torch.Tensor) -> dict:
    """The per-slice record of test_immoco.py:74-93: centre-half crop of the magnitudes -> calmetric2D."""
    H, W = image_gt.shape[-2:]
    c0, c1 = int(H / 4), int(W / 4)
    p = image.abs()[c0:-c0, c1:-c1][None, None]
    g = image_gt.abs()[c0:-c0, c1:-c1][None, None]
    ps, ss, hp, rm = calmetric2D(p, g)
    return {"ssim": ss, "psnr": ps, "haar_psi": hp, "rmse": rm}
