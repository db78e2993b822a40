"""PSNR/RMSE part of the reference's metric path (reference src/utils/evaluate.py:19-47 and the
centre crop of src/test/test_immoco.py:74-85).  SSIM/HaarPSI need `piq`, which is not available,
and are out of scope.  Plain tensor arithmetic, any device, outside the hot path."""
import torch


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
