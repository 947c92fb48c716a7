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
    return v / ((v ** 2).sum() ** 0.5 + eps)


def max_singular_value(W, u=None, Ip=1):
    """Power iteration on the detached weight matrix W [out, in] (SNDiscriminator.py:10-25): returns (sigma [1,1], u)."""
    W = W.detach()
    if u is None:
        u = torch.randn(1, W.size(0), device=W.device, dtype=W.dtype)
    _u = u
    for _ in range(Ip):
        _v = _l2normalize(torch.matmul(_u, W), eps=1e-12)
        _u = _l2normalize(torch.matmul(_v, W.t()), eps=1e-12)
    sigma = torch.matmul(torch.matmul(_v, W.t()), _u.t())
    return sigma, _u


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
        parameter, which restores the reference's behaviour."""
        w_mat = self.weight.view(self.weight.size(0), -1)
        sigma, u = max_singular_value(w_mat, self.u, Ip=self.Ip)
        self.u = u
        with torch.no_grad():
            self.weight.data = self.weight.data / sigma     # in place, cumulative: SNDiscriminator.py:67,91
        return self.weight + 0 if torch.is_grad_enabled() and self.weight.requires_grad else self.weight


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
        return F.linear(input, self._renormalise_(), self.bias)


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
        outs = []
        for t0 in range(T - self.window_size + 1):
            window = input[:, t0:t0 + self.window_size].reshape(B, self.window_size * C, H, W)
            feats = self.conv_layers(window).reshape(B, self.num_sn_linear_in_feats)
            outs.append(self.linear_layer(feats))
        return torch.cat(outs, dim=1)
