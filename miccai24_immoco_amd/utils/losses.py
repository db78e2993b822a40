"""GradientEntropyLoss with the reference's name and semantics
(reference src/utils/losses.py:20-40) on a fused HIP kernel (forward + gradient)."""
import torch

from .. import _lib as L


class _GEFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        L.require_gpu(x, what="GradientEntropyLoss")
        if x.dim() != 2 or x.dtype != torch.complex64:
            raise L.ImmocoError("GradientEntropyLoss expects a 2-D complex64 image")
        x = x.contiguous()
        H, W = x.shape
        loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        grad = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(x.device):
            L.check(L.lib().immoco_ge_loss(L.ptr(x), H, W, 1.0, L.ptr(loss), L.ptr(grad), L.stream_ptr()), "ge_loss")
        ctx.save_for_backward(grad)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g


class GradientEntropyLoss(torch.nn.Module):
    def forward(self, x):
        return _GEFunction.apply(x)
