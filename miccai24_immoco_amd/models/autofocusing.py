"""Autofocusing baseline with the reference's interface (reference src/models/autofocusing.py:8-91,
loop src/test/test_autofocusing.py:66-74): three rigid parameters per motion group, bicubic warp of
the group's masked-k-space image, gradient-entropy loss, torch.optim.Adam(lr=1.0) in the caller.
SURVEY §8(f) rank 4: it reuses the path's operators (IFFT/FFT on rocFFT, GradientEntropyLoss,
line select); the bicubic affine warp and its gradient w.r.t. the affine matrices are HIP kernels.
The small parameter algebra (3 x nM numbers) stays in torch tensor ops on the GPU, including the
reference's `shift[:, 1] = shift[:, 0] + ...` quirk (autofocusing.py:50-53)."""
import torch
import torch.nn as nn

from .. import _lib as L
from ..utils.data_utils import FFT, IFFT


class _AffineBicubic(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, theta):
        L.require_gpu(images, theta, what="affine_bicubic")
        images = images.contiguous()
        theta = theta.contiguous().float()
        n, H, W = images.shape
        xs = torch.linspace(-1, 1, W, device=images.device)
        ys = torch.linspace(-1, 1, H, device=images.device)
        out = torch.empty_like(images)
        L.check(L.lib().immoco_affine_bicubic_fwd(L.ptr(images), L.ptr(theta), L.ptr(xs), L.ptr(ys), n, H, W,
                                                  L.ptr(out), L.stream_ptr()), "affine_bicubic_fwd")
        ctx.save_for_backward(images, theta, xs, ys)
        return out

    @staticmethod
    def backward(ctx, dout):
        images, theta, xs, ys = ctx.saved_tensors
        n, H, W = images.shape
        dtheta = torch.zeros_like(theta)
        dout = dout.contiguous()
        L.check(L.lib().immoco_affine_bicubic_bwd(L.ptr(images), L.ptr(theta), L.ptr(xs), L.ptr(ys), L.ptr(dout),
                                                  n, H, W, L.ptr(dtheta), L.stream_ptr()), "affine_bicubic_bwd")
        return None, dtheta


class Autofocusing(nn.Module):
    def __init__(self, masks):
        super().__init__()
        L.require_gpu(masks, what="Autofocusing(masks)")
        self.num_movements = masks.shape[0]
        dev = masks.device
        self.motion_parameters = nn.ParameterDict(dict(
            rot_vector=nn.Parameter(torch.zeros(self.num_movements, device=dev)),
            x_shifts=nn.Parameter(torch.zeros(self.num_movements, device=dev)),
            y_shifts=nn.Parameter(torch.zeros(self.num_movements, device=dev)),
        ))
        self.device = dev
        self.masks = masks

    def forward(self, ks_input):
        x, num_lines = ks_input.shape
        nM = self.masks.shape[0]
        mf = self.masks.float()
        images = IFFT(ks_input.squeeze().unsqueeze(0) * mf)                      # [nM, H, W]
        angle = torch.deg2rad(self.motion_parameters["rot_vector"])
        cos, sin = torch.cos(angle), torch.sin(angle)
        # rotation_matrix.permute(0, 2, 1) of [[cos, -sin], [sin, cos]]  (autofocusing.py:32-40)
        r00, r01, r10, r11 = cos, sin, -sin, cos
        tx, ty = self.motion_parameters["x_shifts"], self.motion_parameters["y_shifts"]
        s0 = -r00 * tx - r01 * ty
        s1 = s0 + (-r10 * tx - r11 * ty)          # the reference adds shift[:, 0] here (autofocusing.py:50-53)
        # torch_affine[:, :, -1] / (tensor(images[0, 0].shape) * 2 - 1): row 0 by 2H-1, row 1 by 2W-1 (:66-68)
        theta = torch.stack([torch.stack([r00, r01, s0 / (x * 2.0 - 1.0)], -1),
                             torch.stack([r10, r11, s1 / (num_lines * 2.0 - 1.0)], -1)], 1)
        image_2d = _AffineBicubic.apply(images, theta)
        kspace_out = ks_input.squeeze() * (1 - self.masks.sum(0)).float() + (FFT(image_2d) * mf).sum(0)
        return kspace_out
