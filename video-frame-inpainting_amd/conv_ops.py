"""Convolution + bias + activation as the generator uses it (reference: nn.Conv2d / nn.ConvTranspose2d followed by
nn.ReLU, nn.Tanh or nothing).  Inference on the GPU runs the bias-free MIOpen convolution and finishes it with the
single-pass HIP bias+activation kernel of the C ABI (ATen would add the bias and apply the ReLU in two more full passes);
anything that needs autograd, and CPU tensors, take the stock PyTorch ops -- same arithmetic, one rounding per add."""
import torch
import torch.nn.functional as F

from . import _native

_ACT = {None: 0, 'relu': 1, 'tanh': 2}


def _finish(y, bias, act):
    if act == 'relu':
        return torch.relu(y + bias.view(1, -1, 1, 1)) if bias is not None else torch.relu(y)
    if act == 'tanh':
        return torch.tanh(y + bias.view(1, -1, 1, 1)) if bias is not None else torch.tanh(y)
    return y + bias.view(1, -1, 1, 1) if bias is not None else y


def conv_bias_act(x, weight, bias, padding, act):
    """act(conv2d(x, weight, stride 1, padding) + bias), act in {None, 'relu', 'tanh'}."""
    fused = (x.is_cuda and x.dtype == torch.float32 and bias is not None
             and not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or bias.requires_grad)))
    if not fused:
        y = F.conv2d(x, weight, bias, stride=1, padding=padding)
        return torch.relu(y) if act == 'relu' else (torch.tanh(y) if act == 'tanh' else y)
    Co, Ci, kh, kw = weight.shape
    N, _, H, W = x.shape
    thin_in = Ci == 1 and kh == kw and kh in (3, 5) and padding == kh // 2 and W % 4 == 0 and act in (None, 'relu') and Co >= 16
    thin_out = Co == 1 and kh == kw == 3 and padding == 1 and W % 4 == 0 and Ci >= 16
    if thin_in or thin_out:
        # one input or one output channel: no GEMM in it, a stream of the wide tensor (csrc/thin_conv.hip.inc)
        L = _native.lib()
        x, weight = x.contiguous(), weight.contiguous()
        y = torch.empty((N, Co, H, W), dtype=x.dtype, device=x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            if thin_in:
                _native.check(L.tai_conv_cin1_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), N, Co,
                                                      H, W, kh, _ACT[act], stream), 'tai_conv_cin1_forward')
            else:
                _native.check(L.tai_conv_cout1_3x3_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), N,
                                                           Ci, H, W, _ACT[act], stream), 'tai_conv_cout1_3x3_forward')
        return y
    y = F.conv2d(x, weight, None, stride=1, padding=padding)
    if not y.is_contiguous():
        y = y.contiguous()
    N, C, H, W = y.shape
    with torch.cuda.device(y.device):
        _native.check(_native.lib().tai_bias_act_inplace(y.data_ptr(), bias.data_ptr(), N, C, H * W, _ACT[act],
                                                         torch.cuda.current_stream(y.device).cuda_stream),
                      'tai_bias_act_inplace')
    return y
