"""Adaptive separable convolution op: the host-side mirror of the reference's
``src/separable_convolution/SeparableConvolution.py`` (a ``torch.autograd.Function``) on top of the
C-ABI HIP library (include/tai_sepconv.h).

Same call shape ``SeparableConvolution.apply(input, vertical, horizontal, ks)``, same shape and
contiguity asserts (reference :27-33), same ``NotImplementedError`` for CPU tensors (:48-49, :86-87),
same return of ``(grad_input, grad_vertical, grad_horizontal, None)`` (:89).  Differences: outputs are
allocated with ``empty`` instead of ``zero_()`` (the kernels write every element; the reference's
memsets :36,:69-71 are redundant), and launches go to the CURRENT torch stream so the op can be
captured into a hipGraph.
"""
import torch

from . import _native


def _stream_ptr(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class SeparableConvolution(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, vertical, horizontal, ks=51):
        ctx.save_for_backward(input, vertical, horizontal)
        ctx.constant = ks
        B, C, Hin, Win = input.shape
        fsize = min(vertical.size(1), horizontal.size(1))
        Hout = min(vertical.size(2), horizontal.size(2))
        Wout = min(vertical.size(3), horizontal.size(3))
        assert Hin - ks == Hout - 1
        assert Win - ks == Wout - 1
        assert fsize == ks
        assert input.is_contiguous()
        assert vertical.is_contiguous()
        assert horizontal.is_contiguous()
        if not input.is_cuda:
            raise NotImplementedError()  # as the reference: the op exists on the GPU only
        assert vertical.shape == horizontal.shape == (B, ks, Hout, Wout)
        assert input.dtype == vertical.dtype == horizontal.dtype == torch.float32
        output = torch.empty((B, C, Hout, Wout), dtype=input.dtype, device=input.device)
        L = _native.lib()
        with torch.cuda.device(input.device):
            _native.check(L.tai_sepconv_forward(input.data_ptr(), vertical.data_ptr(), horizontal.data_ptr(),
                                                output.data_ptr(), B, C, Hout, Wout, ks, _stream_ptr(input)),
                          'tai_sepconv_forward')
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, vertical, horizontal = ctx.saved_tensors
        ks = ctx.constant
        if not grad_output.is_cuda:
            raise NotImplementedError()
        B, C, _, _ = input.shape
        _, _, Hout, Wout = vertical.shape
        grad_output = grad_output.contiguous()
        need_i, need_v, need_h = ctx.needs_input_grad[:3]
        grad_input = torch.empty_like(input) if need_i else None
        grad_vertical = torch.empty_like(vertical) if need_v else None
        grad_horizontal = torch.empty_like(horizontal) if need_h else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        L = _native.lib()
        with torch.cuda.device(input.device):
            _native.check(L.tai_sepconv_backward(grad_output.data_ptr(), input.data_ptr(), vertical.data_ptr(),
                                                 horizontal.data_ptr(), ptr(grad_input), ptr(grad_vertical),
                                                 ptr(grad_horizontal), B, C, Hout, Wout, ks,
                                                 _stream_ptr(input)), 'tai_sepconv_backward')
        return grad_input, grad_vertical, grad_horizontal, None


def forward_bytes(B, C, H, W, ks):
    """Algorithmic HBM bytes of one forward call."""
    return int(_native.lib().tai_sepconv_forward_bytes(B, C, H, W, ks))


def backward_bytes(B, C, H, W, ks):
    return int(_native.lib().tai_sepconv_backward_bytes(B, C, H, W, ks))


def set_forward_variant(v):
    """Kernel variant selector of the C ABI (0 = automatic); returns the previous value."""
    return int(_native.lib().tai_sepconv_set_forward_variant(int(v)))
