"""Bilinear x2 upsampling (align_corners=True: the reference's torch-0.3.1 ``nn.Upsample(scale_factor=2,
mode='bilinear')``, tai.py:283,337,343) on the HIP kernels of the C ABI (the backward pass as a gather with the forward's
weights: training only; the forward is what the inference path streams eight times per frame).
CPU tensors take the stock ATen path: this op, unlike the separable convolution, exists on the CPU in the reference's
framework too, and host-side tests of the model's control flow run there."""
import torch
import torch.nn.functional as F

from . import _native


class _Upsample2xAlignCorners(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        ctx.in_shape = (B, C, H, W)
        x = x.contiguous()
        out = torch.empty((B, C, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().tai_upsample_bilinear2x_forward(
                x.data_ptr(), out.data_ptr(), B * C, H, W, torch.cuda.current_stream(x.device).cuda_stream),
                'tai_upsample_bilinear2x_forward')
        return out

    @staticmethod
    def backward(ctx, grad_out):
        B, C, H, W = ctx.in_shape
        g = grad_out.contiguous()
        gin = torch.empty((B, C, H, W), dtype=g.dtype, device=g.device)
        with torch.cuda.device(g.device):
            _native.check(_native.lib().tai_upsample_bilinear2x_backward(
                g.data_ptr(), gin.data_ptr(), B * C, H, W, torch.cuda.current_stream(g.device).cuda_stream),
                'tai_upsample_bilinear2x_backward')
        return gin


def upsample2x(x):
    if x.is_cuda and x.dtype == torch.float32:
        return _Upsample2xAlignCorners.apply(x)
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)
