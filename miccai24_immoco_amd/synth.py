"""Synthetic inputs for benchmarks and tests: an analytic complex
phantom and the reference's motion simulator (src/utils/motion_utils.py:7-34,
121-202) re-stated with the same torch RNG call order, so a shared seed gives the
same corruption as the reference.  Harness code: device-generic torch ops,
outside every timed region (no dataset ships with the reference)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def phantom(H: int, W: int, seed: int, n_ellipses: int = 12, device="cpu") -> torch.Tensor:
    """Sum of ellipses with intensities in [0.1, 1] times a smooth quadratic phase (<= pi/4)."""
    g = torch.Generator().manual_seed(int(seed))
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    img = torch.zeros(H, W)
    # a head-like outer ellipse, then random inner structures
    img += 0.35 * ((xx / 0.72) ** 2 + (yy / 0.9) ** 2 <= 1).float()
    for _ in range(n_ellipses - 1):
        cx, cy = ((torch.rand(2, generator=g) - 0.5) * 1.0).tolist()
        a, b = (0.05 + 0.3 * torch.rand(2, generator=g)).tolist()
        th = float(torch.rand(1, generator=g)) * math.pi
        v = 0.1 + 0.9 * float(torch.rand(1, generator=g))
        xr = (xx - cx) * math.cos(th) + (yy - cy) * math.sin(th)
        yr = -(xx - cx) * math.sin(th) + (yy - cy) * math.cos(th)
        inside = ((xr / a) ** 2 + (yr / b) ** 2 <= 1) & ((xx / 0.72) ** 2 + (yy / 0.9) ** 2 <= 1)
        img = img + v * inside.float()
    img = img / img.max()
    phase = (math.pi / 4) * (0.6 * xx * xx - 0.4 * yy * xx + 0.3 * yy) / 1.3
    return (img * torch.exp(1j * phase)).to(torch.complex64).to(device)


def _fft(x):
    return torch.fft.fftshift(torch.fft.fftn(torch.fft.ifftshift(x, dim=(-2, -1)), dim=(-2, -1)), dim=(-2, -1))


def _generate_list(size, n_movements, mingap):
    slack = size - mingap * (n_movements - 1)
    steps = torch.randint(0, slack, (1,))[0]
    inc = torch.hstack([torch.ones((steps,), dtype=torch.long), torch.zeros((n_movements,), dtype=torch.long)])
    inc = inc[torch.randperm(inc.shape[0])]
    locs = torch.argwhere(inc == 0).flatten()
    return torch.cumsum(inc, dim=0)[locs] + mingap * torch.arange(0, n_movements)


def _rand_int(lo, hi):
    r = torch.randint(lo, hi, size=(1,))
    return r + 1 if r == 0 else r


def motion_simulation2D(image_2d: torch.Tensor, n_movements: int):
    """Returns (kspace_corrupt [H,W] c64, line mask [H,W] int64, rotations, translations).
    RNG draws happen on the CPU generator in the reference's order; the image ops run on
    image_2d's device."""
    dev = image_2d.device
    ksp = _fft(image_2d)
    H, W = ksp.shape
    starts = _generate_list(W, n_movements, W // n_movements)
    mask = torch.zeros((H, W), dtype=torch.long, device=dev)
    rots = torch.zeros(n_movements)
    trans = torch.zeros(n_movements, 2)
    for m in range(n_movements):
        sx, sy = _rand_int(-10, 10).item(), _rand_int(-10, 10).item()
        ang = _rand_int(-10, 10)
        a = torch.deg2rad(ang)
        aff = torch.tensor([[torch.cos(a), -torch.sin(a), float(sx)], [torch.sin(a), torch.cos(a), float(sy)]]).view(1, 2, 3)
        aff[:, :, -1] /= W * 2.0 - 1            # reference divides both shifts by 2*W-1 (motion_utils.py:163)
        grid = F.affine_grid(aff, (1, 1, H, W), align_corners=True).to(dev)
        re = F.grid_sample(image_2d.real[None, None], grid, mode="bilinear", padding_mode="border", align_corners=False)
        im = F.grid_sample(image_2d.imag[None, None], grid, mode="bilinear", padding_mode="border", align_corners=False)
        ksp_m = _fft((re + 1j * im).squeeze())
        w0 = int(starts[m])
        w1 = w0 + int(_rand_int(1, 10))
        ksp[..., w0:w1] = ksp_m[..., w0:w1]
        mask[:, w0:w1] = 1
        rots[m] = ang
        trans[m] = torch.tensor([sx, sy])
    return ksp, mask, rots, trans


def make_slice(H: int, W: int, n_movements: int, slice_idx: int, device="cpu"):
    """Seeded synthetic slice: ground truth, corrupted k-space and the voted line flags
    (`mask.sum(0)/H > 0.2`, reference src/test/test_immoco.py:59-61 with the ground-truth mask
    standing in for kLD-Net, whose weights are not available).  On a GPU device the corruption is
    produced by the HIP motion simulator (utils/motion_utils.py, same host RNG draws, validated
    against the reference's golden vectors); on the CPU by the torch restatement above."""
    seed = 1000 + int(slice_idx)
    gt = phantom(H, W, seed, device=device)
    torch.manual_seed(seed)
    if torch.device(device).type == "cuda":
        from .utils.motion_utils import motion_simulation2D as sim_gpu
        ksp, mask, rots, trans = sim_gpu(gt.clone(), n_movements)
    else:
        ksp, mask, rots, trans = motion_simulation2D(gt.clone(), n_movements)
    lines = mask.sum(0).div(H) > 0.2
    return {"gt": gt, "kspace": ksp, "lines": lines, "rotations": rots, "translations": trans}
