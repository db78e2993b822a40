"""kLD-Net: the k-space line detector in front of the IM-MoCo solve.

The reference builds it as ``fastmri.models.Unet(in_chans=2, out_chans=1, chans=32, num_pool_layers=4)``
(``src/models/kld_net.py:4-12``; fastmri==0.3.0 per ``requirements.txt`` - absent here) and uses it in
``src/test/test_immoco.py:16-20,47-61``.  This module restates that network on torch's ROCm ops (MIOpen
convolutions - the once-per-slice caller glue, not the hot path) with the SAME ``state_dict`` layout, so a
checkpoint trained with the reference (``kLDNet.pth``) loads unchanged:

    down_sample_layers.{i}.layers.{0,4}.weight      3x3 conv, no bias   (i = 0 .. pools-1)
    conv.layers.{0,4}.weight                        bottleneck
    up_transpose_conv.{i}.layers.0.weight           2x2 stride-2 transposed conv, no bias
    up_conv.{i}.layers.{0,4}.weight                 (i < pools-1)
    up_conv.{pools-1}.0.layers.{0,4}.weight, up_conv.{pools-1}.1.{weight,bias}   last block + 1x1 conv

Every conv is followed by an affine-free instance norm and LeakyReLU(0.2); Dropout2d(drop_prob) follows each
activation of the 3x3 blocks (identity in eval mode and for drop_prob = 0, the reference's setting).
Pinned by golden vectors generated with the reference's own U-Net source (``src/models/unet.py:16-187`` with
``batchnorm=nn.InstanceNorm2d``, which is the fastmri architecture): tests/golden/kld_net.npz.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from ..utils.data_utils import IFFT
from ..utils.motion_utils import extract_movement_groups

_SLOPE = 0.2


def _norm_act(x: torch.Tensor) -> torch.Tensor:
    return F.leaky_relu(F.instance_norm(x), _SLOPE)


class _Slots(nn.Module):
    """Holds child modules under explicit numeric names (keeps the checkpoint key layout)."""

    def __init__(self, **mods):
        super().__init__()
        for k, m in mods.items():
            self.add_module(k.lstrip("_"), m)


class _DoubleConv(nn.Module):
    def __init__(self, cin: int, cout: int, drop_prob: float):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.layers = _Slots(_0=nn.Conv2d(cin, cout, 3, padding=1, bias=False),
                             _4=nn.Conv2d(cout, cout, 3, padding=1, bias=False))

    def forward(self, x):
        for name in ("0", "4"):
            x = _norm_act(getattr(self.layers, name)(x))
            if self.drop_prob > 0.0:
                x = F.dropout2d(x, self.drop_prob, self.training)
        return x


class _UpConv(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.layers = _Slots(_0=nn.ConvTranspose2d(cin, cout, 2, stride=2, bias=False))

    def forward(self, x):
        return _norm_act(getattr(self.layers, "0")(x))


class _Head(nn.Module):
    """Last decoder stage: double conv followed by the 1x1 output conv (keys ``0.layers.*`` / ``1.*``)."""

    def __init__(self, cin: int, cout: int, n_out: int, drop_prob: float):
        super().__init__()
        self.add_module("0", _DoubleConv(cin, cout, drop_prob))
        self.add_module("1", nn.Conv2d(cout, n_out, 1))

    def forward(self, x):
        return getattr(self, "1")(getattr(self, "0")(x))


class Unet(nn.Module):
    """fastMRI U-Net (encoder: double conv + 2x2 average pool; decoder: transposed conv + skip concat)."""

    def __init__(self, in_chans: int, out_chans: int, chans: int = 32, num_pool_layers: int = 4,
                 drop_prob: float = 0.0):
        super().__init__()
        if num_pool_layers < 1:
            raise ValueError("num_pool_layers must be >= 1")
        self.in_chans, self.out_chans, self.chans = in_chans, out_chans, chans
        self.num_pool_layers, self.drop_prob = num_pool_layers, drop_prob
        widths = [chans << i for i in range(num_pool_layers)]
        self.down_sample_layers = nn.ModuleList(
            _DoubleConv(cin, c, drop_prob) for cin, c in zip([in_chans] + widths[:-1], widths))
        self.conv = _DoubleConv(widths[-1], 2 * widths[-1], drop_prob)
        self.up_conv = nn.ModuleList(
            [_DoubleConv(2 * c, c, drop_prob) for c in reversed(widths[1:])]
            + [_Head(2 * widths[0], widths[0], out_chans, drop_prob)])
        self.up_transpose_conv = nn.ModuleList(_UpConv(2 * c, c) for c in reversed(widths))

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        if image.dim() != 4 or image.shape[1] != self.in_chans:
            raise ValueError(f"expected (N, {self.in_chans}, H, W), got {tuple(image.shape)}")
        skips, x = [], image
        for enc in self.down_sample_layers:
            x = enc(x)
            skips.append(x)
            x = F.avg_pool2d(x, 2)
        x = self.conv(x)
        for up, dec in zip(self.up_transpose_conv, self.up_conv):
            skip = skips.pop()
            x = up(x)
            # odd sizes lose a row/column in the pooling: reflect-pad right/bottom back to the skip's size
            dh, dw = skip.shape[-2] - x.shape[-2], skip.shape[-1] - x.shape[-1]
            if dh or dw:
                x = F.pad(x, [0, 1 if dw else 0, 0, 1 if dh else 0], "reflect")
            x = dec(torch.cat([x, skip], dim=1))
        return x


def get_unet(in_chans: int, out_chans: int, chans: int, num_pool_layers: int, drop_prob: float, **kwargs):
    """Same factory as the reference (``src/models/kld_net.py:4-12``)."""
    return Unet(in_chans=in_chans, out_chans=out_chans, chans=chans, num_pool_layers=num_pool_layers,
                drop_prob=drop_prob, **kwargs)


# ----------------------------------------------------------------------------------------------
# caller glue of src/test/test_immoco.py:47-61
@torch.no_grad()
def detect_line_mask(net: nn.Module, kspace: torch.Tensor) -> torch.Tensor:
    """Per-pixel corrupted-line prediction of kLD-Net for ONE slice: ``kspace [H, W] c64`` ->
    bool ``[H, W]``.  Input scaling as the reference: k-space divided by the std of the magnitude image,
    (re, im) as two channels; ``sigmoid(.) > 0.5`` (test_immoco.py:50-58)."""
    if kspace.dim() != 2 or not kspace.is_complex():
        raise ValueError("kspace must be a complex [H, W] tensor")
    k = kspace[None, None]
    scale = IFFT(k).abs().std()
    x = torch.view_as_real(k / scale).squeeze(1).permute(0, 3, 1, 2).contiguous()
    return (net(x).sigmoid() > 0.5)[0, 0]


def vote_lines(mask: torch.Tensor, threshold: float = 0.2) -> torch.Tensor:
    """Column vote ``mask.sum(0) / H > 0.2`` (test_immoco.py:59-61): bool ``[W]``."""
    return mask.sum(0).div(mask.shape[0]) > threshold


def detect_movement_groups(net: nn.Module, kspace: torch.Tensor, threshold: float = 0.2) -> torch.Tensor:
    """kLD-Net -> vote -> ``extract_movement_groups(..., make_list=True)``: ``[nM, H, W] int64`` one-hot
    masks on the k-space's device, ready for ``imcoco_motion_correction``."""
    lines = vote_lines(detect_line_mask(net, kspace), threshold)
    return extract_movement_groups(lines, make_list=True).to(kspace.device)
