"""The reference's evaluation driver (src/test/test_immoco.py:33-93) as a function: for every slice
kLD-Net detects the corrupted lines, the lines are voted into movement groups, the HIP solver corrects the
slice, and the four metrics are computed on the device.

The reference runs this as a script over ``Dataset/Brain/t2/test_files/_test_data_{light,heavy}.pth`` with a
trained ``kLDNet.pth`` (neither ships with the repository); here the same steps are exposed for tensors the
caller already holds.  Nothing is copied back to the host except the final metric scalars on request.
"""
from __future__ import annotations

from typing import Optional

import torch

from .models import kld_net
from .models.immoco import imcoco_motion_correction
from .utils.evaluate import slice_metrics


def correct_slice(kspace: torch.Tensor, net: Optional[torch.nn.Module] = None, masks: Optional[torch.Tensor] = None,
                  iters: int = 200, learning_rate: float = 1e-2, lambda_ge: float = 1e-2, **solver_kwargs):
    """One pass of test_immoco.py:47-70 for ``kspace [H, W] c64`` on the GPU.  Movement groups come from
    ``net`` (kLD-Net, test_immoco.py:50-61) unless ``masks [nM, H, W]`` are given (e.g. ground-truth bands).
    Returns ``(refined_image [H, W] c64, masks)``."""
    if (net is None) == (masks is None):
        raise ValueError("pass exactly one of net (kLD-Net) or masks")
    if masks is None:
        masks = kld_net.detect_movement_groups(net, kspace)
    image, _ = imcoco_motion_correction(kspace, masks, iters=iters, learning_rate=learning_rate,
                                        lambda_ge=lambda_ge, debug=False, **solver_kwargs)
    return image, masks


def evaluate_slices(kspaces: torch.Tensor, images_gt: torch.Tensor, net: Optional[torch.nn.Module] = None,
                    masks_list=None, iters: int = 200, to_host: bool = True, **solver_kwargs):
    """The loop of test_immoco.py:45-93 over ``kspaces [B, H, W] c64`` with ground truth ``images_gt
    [B, H, W]``: list of ``{"ssim", "psnr", "haar_psi", "rmse"}`` records (floats when ``to_host``, else
    0-d device tensors) and the corrected images ``[B, H, W] c64``."""
    if kspaces.dim() != 3 or images_gt.shape != kspaces.shape:
        raise ValueError("kspaces and images_gt must both be [B, H, W]")
    if net is not None:
        net.eval()
    records, images = [], []
    for i in range(kspaces.shape[0]):
        image, _ = correct_slice(kspaces[i], net=net, masks=None if masks_list is None else masks_list[i],
                                 iters=iters, **solver_kwargs)
        rec = slice_metrics(image.detach(), images_gt[i].to(image.device))
        records.append({k: float(v) for k, v in rec.items()} if to_host else rec)
        images.append(image.detach())
    return records, torch.stack(images)
