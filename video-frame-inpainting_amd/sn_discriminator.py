"""Spectral-norm sliding-window video discriminator (reference src/discriminators/SNDiscriminator.py:10-159).

The reference's spectral normalisation is NOT a re-parameterisation: every forward runs ``Ip`` power iterations from a
persistent vector ``u`` (drawn N(0,1) on first use, :16-19) and then overwrites ``weight.data <- weight.data / sigma``
in place (:67, :91) -- cumulative, every call, 13 windows x 3 discriminator forwards per training step at K=T=F=5.
That behaviour is part of the training dynamics and is kept exactly.  ``u`` stays out of the state dict as in the
reference (a plain attribute there; a non-persistent buffer here so ``.to(device)`` and the data-parallel broadcast
of parallel.py see it).
"""
from math import floor

import torch
import torch.nn as nn
import torch.nn.functional as F


def _l2normalize(v, eps=1e-12):
    return v / (torch.linalg.vector_norm(v) + eps)


def max_singular_value(W, u=None, Ip=1):
    """Power iteration on the detached weight matrix W [out, in] (SNDiscriminator.py:10-25): returns (sigma [1,1], u).

    Same arithmetic as the reference, issued as matrix-vector products: ``matmul`` of a [1, out] row with W goes to the
    GEMM library, which takes 90 us for the [512, 4096] layer (a 16-row macro tile for one row of work; 312 such calls
    were 28 ms of a 480 ms training step), ``mv`` streams W once (tools/sn_bench.py).  ``(v ** 2).sum() ** 0.5`` is one
    ``vector_norm`` kernel instead of three, and sigma reuses the last ``v W^T`` product instead of recomputing it."""
    W = W.detach()
    if u is None:
        u = torch.randn(1, W.size(0), device=W.device, dtype=W.dtype)
    _u = u.reshape(-1)
    Wt = W.t()
    for _ in range(Ip):
        _v = _l2normalize(torch.mv(Wt, _u), eps=1e-12)
        t = torch.mv(W, _v)
        _u = _l2normalize(t, eps=1e-12)
    sigma = torch.dot(t, _u)
    return sigma.view(1, 1), _u.view(1, -1)


class _SpectralNormalised(object):
    def _init_sn(self, Ip):
        self.Ip = Ip
        self.register_buffer('u', None, persistent=False)

    def _renormalise_(self):
        """Renormalises the parameter in place and returns the weight to convolve with.  Under torch 0.3.1 autograd kept
        the weight TENSOR of the moment for the backward pass, so each window back-propagates through the weights it
        was evaluated with, although 12 later windows (and later forwards) have rescaled the parameter since.  Modern
        autograd keeps a leaf parameter by reference and would back-propagate through the LATEST ``.data``; the
        returned ``weight + 0`` is a value of the moment with its own storage whose gradient still accumulates in the
        parameter, which restores the reference's behaviour.

        On the GPU the power iteration and the division are ``tai_sn_power_iteration`` (2 Ip + 1 launches instead of
        ~30 ATen kernels; csrc/spectral_norm.hip.inc); host tensors take the ATen form below."""
        weight = self.weight
        if weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous():
            self._renormalise_native_()
        else:
            w_mat = weight.view(weight.size(0), -1)
            sigma, u = max_singular_value(w_mat, self.u, Ip=self.Ip)
            self.u = u
            with torch.no_grad():
                weight.data = weight.data / sigma     # in place, cumulative: SNDiscriminator.py:67,91
        return weight + 0 if torch.is_grad_enabled() and weight.requires_grad else weight

    def _renormalise_native_(self, sigma_out=None):
        """One renormalisation on the GPU; sigma goes to ``sigma_out`` (a one-element device view) or to the layer's own slot."""
        from . import _native
        weight = self.weight
        out_rows = weight.size(0)
        in_cols = weight.numel() // out_rows
        if self.u is None:
            self.u = torch.randn(1, out_rows, device=weight.device, dtype=weight.dtype)     # SNDiscriminator.py:16-19
        u = self.u
        if not (u.is_cuda and u.device == weight.device and u.dtype == torch.float32 and u.is_contiguous() and u.numel() == out_rows):
            raise ValueError('spectral-norm vector u must be a contiguous fp32 [1, %d] tensor on %s' % (out_rows, weight.device))
        scratch = getattr(self, '_sn_scratch', None)
        if scratch is None or scratch.device != weight.device or scratch.numel() != in_cols + out_rows + 1:
            scratch = self._sn_scratch = torch.empty(in_cols + out_rows + 1, device=weight.device, dtype=torch.float32)
        sigma = scratch[-1:] if sigma_out is None else sigma_out
        with torch.cuda.device(weight.device):
            _native.check(_native.lib().tai_sn_power_iteration(
                weight.data_ptr(), u.data_ptr(), scratch.data_ptr(), sigma.data_ptr(), out_rows, in_cols, int(self.Ip),
                torch.cuda.current_stream(weight.device).cuda_stream), 'tai_sn_power_iteration')

    @property
    def last_sigma(self):
        """sigma of the most recent single GPU renormalisation (a device scalar), for tests and diagnostics."""
        return self._sn_scratch[-1:] if getattr(self, '_sn_scratch', None) is not None else None

    def renormalise_sequence_(self, count):
        """``count`` renormalisations in a row (what ``count`` consecutive forwards do to the parameter; none of them
        depends on the data).  Returns (w0, inv_scale): the weight before the first one, and inv_scale[t] = 1 / (sigma_1 ...
        sigma_{t+1}), so that the weight forward t+1 of the reference convolves with is w0 * inv_scale[t]."""
        w0 = self.weight.detach().clone()
        sigmas = torch.empty(count, device=self.weight.device, dtype=torch.float32)
        for t in range(count):
            self._renormalise_native_(sigmas[t:t + 1])
        return w0, torch.cumprod(sigmas, 0).reciprocal_()


def _conv_ops():
    from . import conv_ops
    return conv_ops


# The discriminator's layers are 4x4, stride 2, padding 1 (SNDiscriminator.py:113-133).  On planes cut into 2 x 2 pixel blocks
# (space-to-depth: channel (c, ry, rx) of block (qy, qx) = pixel (2 qy + ry, 2 qx + rx) of channel c) such a layer is a 3x3, stride-1,
# padding-1 layer over 4 C channels: output o reads pixels 2 o - 1 ... 2 o + 2 = (block o - 1, r = 1), (o, 0), (o, 1), (o + 1, 0) per
# dimension, so tap t of the 4-tap filter sits at (T, r) with 2 T + r = t + 1 and the two other (T, r) slots hold zeros.  That 3x3 layer
# runs on the in-tree Winograd kernels -- F(4x4, 3x3) hands the fp32 MFMA 36 / 16 x 4 C = 9 C multiply-adds per output and filter
# where the direct form has 16 C -- forward, input gradient (the same kernels with the weight transposed) and weight gradient
# (tai_conv3x3_wino_wrw, then folded back to [K, C, 4, 4]); bit-reproducible, which MIOpen's backward kernels are not.
def _s2d_applies(x, w, stride, padding):
    return (x.is_cuda and x.dtype == torch.float32 and tuple(w.shape[2:]) == (4, 4) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0)


def _s2d_weight(w):
    """[K, C, 4, 4] -> the 3x3 weight [K, 4 C, 3, 3] over space-to-depth channels (c, ry, rx) (F.pixel_unshuffle's order)."""
    K, C = w.shape[0], w.shape[1]
    return F.pad(w, (1, 1, 1, 1)).view(K, C, 3, 2, 3, 2).permute(0, 1, 3, 5, 2, 4).reshape(K, 4 * C, 3, 3)


def _s2d_weight_grad(g3, C):
    """Gradient of the 3x3 form [K, 4 C, 3, 3] -> gradient of the 4x4 filter [K, C, 4, 4] (the structurally-zero slots dropped)."""
    K = g3.shape[0]
    return g3.view(K, C, 2, 2, 3, 3).permute(0, 1, 4, 2, 5, 3).reshape(K, C, 6, 6)[:, :, 1:5, 1:5].contiguous()


class _WindowScaledConv(torch.autograd.Function):
    """y[t] = conv(x[t], w0) * inv_scale[t] + bias for the ``nw`` window groups stacked along the batch: the convolution
    of window t with the weight the reference holds at that moment, w0 * inv_scale[t] (SNDiscriminator.py:60-68), as ONE
    convolution over all windows.  Gradients as the reference's autograd produces them: every window differentiates with
    respect to ITS weight tensor and the results accumulate in the parameter (so the weight gradient takes the output
    gradient unscaled), the input gradient goes through w0 * inv_scale[t]."""

    @staticmethod
    def forward(ctx, x, weight, w0, bias, inv_scale, nw, stride, padding):
        y = F.conv2d(x, w0, None, stride, padding)
        N = y.shape[0]
        y = (y.view(nw, N // nw, -1) * inv_scale.view(nw, 1, 1)).view_as(y)
        if bias is not None:
            y = y + bias.view(1, -1, 1, 1)
        ctx.save_for_backward(x, w0, inv_scale)
        ctx.cfg = (nw, stride, padding, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w0, inv_scale = ctx.saved_tensors
        nw, stride, padding, has_bias = ctx.cfg
        g = g.contiguous()
        N, Co = g.shape[0], g.shape[1]
        gx = gw = gb = None
        conv_bwd = torch.ops.aten.convolution_backward
        if ctx.needs_input_grad[1]:
            gw = conv_bwd(g, x, w0, [Co], list(stride), list(padding), [1, 1], False, [0, 0], 1, [False, True, False])[1]
        if ctx.needs_input_grad[0]:
            gs = (g.view(nw, N // nw, -1) * inv_scale.view(nw, 1, 1)).view_as(g)
            gx = conv_bwd(gs, x, w0, [Co], list(stride), list(padding), [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if has_bias and ctx.needs_input_grad[3]:
            gb = g.sum((0, 2, 3))
        return gx, gw, None, gb, None, None, None, None


class _WindowScaledConvLReLU(torch.autograd.Function):
    """_WindowScaledConv followed by LeakyReLU(slope), the element-wise tail (per-window factor, bias, activation) in one
    pass forwards and one backwards (tai_window_scale_bias_lrelu / _backward) instead of three and two."""

    @staticmethod
    def forward(ctx, x, weight, w0, bias, inv_scale, nw, stride, padding, slope):
        from . import _native
        xs = w3 = y = None
        if _s2d_applies(x, w0, stride, padding):
            xs, w3 = F.pixel_unshuffle(x, 2), _s2d_weight(w0)
            y = _conv_ops().wino_conv3x3_plain(xs, w3)
        if y is None:
            xs = w3 = None
            y = F.conv2d(x, w0, None, stride, padding).contiguous()
        N, Co, H, W = y.shape
        with torch.cuda.device(y.device):
            _native.check(_native.lib().tai_window_scale_bias_lrelu(
                y.data_ptr(), bias.data_ptr(), inv_scale.data_ptr(), nw, N // nw, Co, H * W, float(slope),
                torch.cuda.current_stream(y.device).cuda_stream), 'tai_window_scale_bias_lrelu')
        ctx.s2d = xs is not None
        if ctx.s2d:
            ctx.save_for_backward(x, w0, inv_scale, y, xs, w3)
        else:
            ctx.save_for_backward(x, w0, inv_scale, y)
        ctx.cfg = (nw, stride, padding, float(slope))
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _native
        x, w0, inv_scale, y = ctx.saved_tensors[:4]
        xs, w3 = ctx.saved_tensors[4:] if ctx.s2d else (None, None)
        nw, stride, padding, slope = ctx.cfg
        g = g.contiguous()
        N, Co, H, W = g.shape
        gz, gs = torch.empty_like(g), torch.empty_like(g)
        with torch.cuda.device(g.device):
            _native.check(_native.lib().tai_window_scale_lrelu_backward(
                g.data_ptr(), y.data_ptr(), inv_scale.data_ptr(), gz.data_ptr(), gs.data_ptr(), nw, N // nw, Co, H * W, slope,
                torch.cuda.current_stream(g.device).cuda_stream), 'tai_window_scale_lrelu_backward')
        gx = gw = gb = None
        conv_bwd = torch.ops.aten.convolution_backward
        if ctx.needs_input_grad[1]:
            if ctx.s2d:                 # the 3x3 layer's Winograd-domain weight gradient (and the bias gradient with it), folded back to 4x4
                both = _conv_ops().wino_weight_grad(xs, gz, with_bias=bool(ctx.needs_input_grad[3]))
                if both is not None:
                    g3, gb = both if ctx.needs_input_grad[3] else (both, None)
                    gw = _s2d_weight_grad(g3, x.shape[1])
            if gw is None:
                gw = conv_bwd(gz, x, w0, [Co], list(stride), list(padding), [1, 1], False, [0, 0], 1, [False, True, False])[1]
        if ctx.needs_input_grad[0]:
            gxs = _conv_ops().wino_conv3x3_plain(gs, w3, transposed=True) if ctx.s2d else None
            if gxs is not None:
                gx = F.pixel_shuffle(gxs, 2)
            else:
                gx = conv_bwd(gs, x, w0, [Co], list(stride), list(padding), [1, 1], False, [0, 0], 1, [True, False, False])[0]
        if ctx.needs_input_grad[3] and gb is None:
            gb = gz.sum((0, 2, 3))
        return gx, gw, None, gb, None, None, None, None, None


class _WindowScaledLinear(torch.autograd.Function):
    """The same for the one-logit linear layer: y[t] = feats[t] @ w0^T * inv_scale[t] + bias."""

    @staticmethod
    def forward(ctx, x, weight, w0, bias, inv_scale, nw):
        y = torch.mv(x, w0.view(-1))
        y = (y.view(nw, -1) * inv_scale.view(nw, 1)).reshape(-1, 1)
        if bias is not None:
            y = y + bias
        ctx.save_for_backward(x, w0, inv_scale)
        ctx.nw = nw
        return y

    @staticmethod
    def backward(ctx, g):
        x, w0, inv_scale = ctx.saved_tensors
        nw = ctx.nw
        g = g.reshape(-1)
        gx = gw = gb = None
        if ctx.needs_input_grad[1]:
            gw = torch.mv(x.t(), g).view_as(w0)
        if ctx.needs_input_grad[0]:
            gs = (g.view(nw, -1) * inv_scale.view(nw, 1)).reshape(-1)
            gx = torch.outer(gs, w0.view(-1))
        if ctx.needs_input_grad[3]:
            gb = g.sum().view(1)
        return gx, gw, None, gb, None, None


class SNConv2d(nn.Conv2d, _SpectralNormalised):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True, Ip=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self._init_sn(Ip)

    def forward(self, input):
        return F.conv2d(input, self._renormalise_(), self.bias, self.stride, self.padding, self.dilation, self.groups)


class SNLinear(nn.Linear, _SpectralNormalised):
    def __init__(self, in_features, out_features, bias=True, Ip=1):
        super().__init__(in_features, out_features, bias)
        self._init_sn(Ip)

    def forward(self, input):
        weight = self._renormalise_()
        if self.out_features == 1 and input.dim() == 2 and self.bias is not None:
            # one logit per clip: a matrix-vector product (the GEMM library runs this 1-column shape at 230 us per call)
            return torch.addmv(self.bias, input, weight.view(-1)).unsqueeze(1)
        return F.linear(input, weight, self.bias)


class SNDiscriminator(nn.Module):
    """4 x [SNConv2d 4x4 stride 2 pad 1 + LeakyReLU(0.2)] (window*C -> df -> 2df -> 4df -> 8df), SNLinear -> 1 logit,
    slid over every window of ``window_size`` consecutive frames (SNDiscriminator.py:95-159).
    Parameter names: conv_layers.{0,2,4,6}, linear_layer."""

    def __init__(self, img_size, c_dim, window_size, df_dim, Ip):
        super().__init__()
        self.window_size = window_size
        h, w = img_size[0], img_size[1]
        layers, cin = [], c_dim * window_size
        for mult in (1, 2, 4, 8):
            layers += [SNConv2d(cin, df_dim * mult, 4, stride=2, padding=1, Ip=Ip), nn.LeakyReLU(0.2)]
            cin = df_dim * mult
            h = floor((h + 2 * 1 - 4) / 2 + 1)
            w = floor((w + 2 * 1 - 4) / 2 + 1)
        self.conv_layers = nn.Sequential(*layers)
        self.num_sn_linear_in_feats = int(h * w * df_dim * 8)
        self.linear_layer = SNLinear(self.num_sn_linear_in_feats, 1, Ip=1)

    def forward(self, input):
        """input [B, T, C, H, W] -> logits [B, T - window_size + 1]."""
        B, T, C, H, W = input.shape
        nw = T - self.window_size + 1
        if input.is_cuda and input.dtype == torch.float32 and self.linear_layer.out_features == 1 and nw >= 1 and \
                all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in self.parameters()):
            return self._forward_windows_batched(input, nw)
        outs = []
        for t0 in range(nw):
            window = input[:, t0:t0 + self.window_size].reshape(B, self.window_size * C, H, W)
            feats = self.conv_layers(window).reshape(B, self.num_sn_linear_in_feats)
            outs.append(self.linear_layer(feats))
        return torch.cat(outs, dim=1)

    def _forward_windows_batched(self, input, nw):
        """All windows in one pass.  The reference evaluates the windows one after the other and every evaluation divides
        every layer's weight by its spectral norm again (:60-68, :84-92, :140-159): window t sees the weights renormalised
        t times.  Those renormalisations do not depend on the data and only rescale the weight, so they are done up front
        (same kernels, same order, the parameter ends in the same state) and the 13 x 5 small convolutions of a K=T=F=5
        clip become 5 convolutions over 13 B images with one factor per window (_WindowScaledConv)."""
        B, T, C, H, W = input.shape
        ws = self.window_size
        x = torch.stack([input[:, t0:t0 + ws].reshape(B, ws * C, H, W) for t0 in range(nw)], dim=0).view(nw * B, ws * C, H, W)
        layers = list(self.conv_layers)
        i = 0
        while i < len(layers):
            layer = layers[i]
            if isinstance(layer, SNConv2d):
                w0, inv_scale = layer.renormalise_sequence_(nw)
                nxt = layers[i + 1] if i + 1 < len(layers) else None
                oh = (x.shape[2] + 2 * layer.padding[0] - layer.kernel_size[0]) // layer.stride[0] + 1
                ow = (x.shape[3] + 2 * layer.padding[1] - layer.kernel_size[1]) // layer.stride[1] + 1
                if isinstance(nxt, nn.LeakyReLU) and layer.bias is not None and (oh * ow) % 4 == 0:
                    x = _WindowScaledConvLReLU.apply(x, layer.weight, w0, layer.bias, inv_scale, nw, layer.stride, layer.padding,
                                                     nxt.negative_slope)
                    i += 2
                    continue
                x = _WindowScaledConv.apply(x, layer.weight, w0, layer.bias, inv_scale, nw, layer.stride, layer.padding)
            else:
                x = layer(x)
            i += 1
        w0, inv_scale = self.linear_layer.renormalise_sequence_(nw)
        logits = _WindowScaledLinear.apply(x.reshape(nw * B, self.num_sn_linear_in_feats), self.linear_layer.weight, w0,
                                           self.linear_layer.bias, inv_scale, nw)
        return logits.view(nw, B).t().contiguous()
