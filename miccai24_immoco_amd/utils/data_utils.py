"""Centred FFTs with the reference's names and semantics
(reference src/utils/data_utils.py:29-34) on rocFFT through the C-ABI."""
import torch

from .. import _lib as L


def _fft2c(x: torch.Tensor, mode: int) -> torch.Tensor:
    L.require_gpu(x, what="FFT/IFFT")
    if x.dim() < 2:
        raise L.ImmocoError("FFT/IFFT need at least 2 dims")
    if not x.is_complex():
        x = x.to(torch.complex64)
    if x.dtype != torch.complex64:
        raise L.ImmocoError(f"FFT/IFFT support complex64 only (got {x.dtype})")
    x = x.contiguous()
    H, W = x.shape[-2:]
    batch = x.numel() // (H * W) if x.numel() else 0
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        L.check(L.lib().immoco_fft2c(L.ptr(x), L.ptr(out), batch, H, W, mode, L.stream_ptr()), "fft2c")
    return out


class _CenteredFFT(torch.autograd.Function):
    """mode 0: FFT (adjoint = mode 2); mode 1: IFFT (adjoint = mode 3: the IFFT's own shifts around the
    forward transform, / (HW) - equal to FFT / (HW) for even sizes only)."""

    @staticmethod
    def forward(ctx, x, mode):
        ctx.mode = mode
        return _fft2c(x, mode)

    @staticmethod
    def backward(ctx, g):
        if ctx.mode == 0:
            return _fft2c(g, 2), None
        return _fft2c(g, 3), None


def FFT(x):
    """fftshift(fftn(ifftshift(x, (-2,-1)), (-2,-1)), (-2,-1)) — data_utils.py:29-30."""
    return _CenteredFFT.apply(x, 0)


def IFFT(x):
    """ifftshift(ifftn(fftshift(x, (-2,-1)), (-2,-1)), (-2,-1)) — data_utils.py:33-34."""
    return _CenteredFFT.apply(x, 1)
