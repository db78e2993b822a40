"""Synthetic inputs for benchmarks and tests: an analytic complex
phantom pushed through the GPU motion simulator (utils/motion_utils.py, the HIP
restatement of src/utils/motion_utils.py:7-34,121-202 with the reference's RNG call
order).  Harness code outside every timed region (no dataset ships with the
reference).  The CPU counterpart used by the oracle side of the tests lives in
oracle/synth_cpu.py."""
from __future__ import annotations

import math

import torch


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


def make_slice(H: int, W: int, n_movements: int, slice_idx: int, device="cuda"):
    """Seeded synthetic slice: ground truth, corrupted k-space and the voted line flags
    (`mask.sum(0)/H > 0.2`, reference src/test/test_immoco.py:59-61 with the ground-truth mask
    standing in for kLD-Net, whose weights are not available).  The corruption is produced by the HIP
    motion simulator (validated against the reference's golden vectors); GPU only."""
    if torch.device(device).type != "cuda":
        from ._lib import ImmocoError
        raise ImmocoError("synth.make_slice runs the HIP motion simulator: pass a cuda device "
                          "(the CPU generator of the tests is oracle/synth_cpu.py)")
    from .utils.motion_utils import motion_simulation2D as sim_gpu
    seed = 1000 + int(slice_idx)
    gt = phantom(H, W, seed, device=device)
    torch.manual_seed(seed)
    ksp, mask, rots, trans = sim_gpu(gt.clone(), n_movements)
    lines = mask.sum(0).div(H) > 0.2
    return {"gt": gt, "kspace": ksp, "lines": lines, "rotations": rots, "translations": trans}
