"""CPU restatement of the reference's bi-TAI forward pass -- TEST INFRASTRUCTURE ONLY.

The reference has no CPU path (src/options/options.py:61 asserts CUDA, src/models/tai/tai.py:72,216
hard-code ``.cuda()``, the sepconv op raises NotImplementedError on CPU), so this file restates its
arithmetic as plain functions over a ``state_dict`` (the reference's key schema) using PyTorch CPU
ops for the convolutions and ``oracle.sepconv_oracle`` (the C restatement of the CUDA kernels) for
the separable convolution.  It is deliberately written as flat functions of tensors, sharing no
code with the product package.

Pinned by tests/golden/*.npz, which were produced by the reference's own classes imported in the
build container (tests/golden/make_golden.py): every MC-Net primitive, the kernel-network building
blocks, and whole-model ``MCNet.forward`` / ``TAIFillInModel.forward`` runs in which the only
substituted piece is the (CPU-less) sepconv call.

Version trap restated on purpose: torch 0.3.1's ``nn.Upsample(scale_factor=2, mode='bilinear')``
(tai.py:283,337,343) is what modern PyTorch calls ``align_corners=True``.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import sepconv_oracle


# ----------------------------------------------------------------------------- util.py:22-41
def inverse_transform(x):
    """util.py:22-23"""
    return (x + 1.) / 2


def bgr2gray(img):
    """util.py:30-34 -- img [B,3,H,W] in BGR order."""
    g = 0.1140 * img[:, 0] + 0.5870 * img[:, 1] + 0.2989 * img[:, 2]
    return g.unsqueeze(1)


def bgr2gray_batched(img):
    """util.py:37-41 -- img [B,T,3,H,W]."""
    g = 0.1140 * img[:, :, 0] + 0.5870 * img[:, :, 1] + 0.2989 * img[:, :, 2]
    return g.unsqueeze(2)


# ----------------------------------------------------------------------------- helpers
def _conv(sd, key, x, pad):
    return F.conv2d(x, sd[key + '.weight'], sd[key + '.bias'], stride=1, padding=pad)


def _convt(sd, key, x):
    # nn.ConvTranspose2d(cin, cout, 3, padding=1): weight is [cin, cout, 3, 3]   (mcnet.py:203-225)
    return F.conv_transpose2d(x, sd[key + '.weight'], sd[key + '.bias'], stride=1, padding=1)


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True)


# ----------------------------------------------------------------------------- mcnet.py:14-294
def motion_enc(sd, p, diff):
    """MotionEnc.forward, mcnet.py:47-60 (layers :28-45)."""
    c1 = F.relu(_conv(sd, p + 'dyn_conv1.0', diff, 2))
    c2 = F.relu(_conv(sd, p + 'dyn_conv2.1', F.max_pool2d(c1, 2), 2))
    c3 = F.relu(_conv(sd, p + 'dyn_conv3.1', F.max_pool2d(c2, 2), 3))
    return F.max_pool2d(c3, 2), [c1, c2, c3]


def content_enc(sd, p, raw):
    """ContentEnc.forward, mcnet.py:106-119 (layers :79-104)."""
    a = F.relu(_conv(sd, p + 'cont_conv1.0', raw, 1))
    c1 = F.relu(_conv(sd, p + 'cont_conv1.2', a, 1))
    a = F.relu(_conv(sd, p + 'cont_conv2.1', F.max_pool2d(c1, 2), 1))
    c2 = F.relu(_conv(sd, p + 'cont_conv2.3', a, 1))
    a = F.relu(_conv(sd, p + 'cont_conv3.1', F.max_pool2d(c2, 2), 1))
    a = F.relu(_conv(sd, p + 'cont_conv3.3', a, 1))
    c3 = F.relu(_conv(sd, p + 'cont_conv3.5', a, 1))
    return F.max_pool2d(c3, 2), [c1, c2, c3]


def comb_layers(sd, p, h_dyn, h_cont):
    """CombLayers.forward, mcnet.py:146-153."""
    x = torch.cat((h_dyn, h_cont), dim=1)
    x = F.relu(_conv(sd, p + 'h_comb.0', x, 1))
    x = F.relu(_conv(sd, p + 'h_comb.2', x, 1))
    return F.relu(_conv(sd, p + 'h_comb.4', x, 1))


def residual(sd, p, a, b):
    """Residual.forward, mcnet.py:178-185 -- no activation after the second conv (:172-176)."""
    x = torch.cat((a, b), dim=1)
    return _conv(sd, p + 'res.2', F.relu(_conv(sd, p + 'res.0', x, 1)), 1)


def fixed_unpooling(x):
    """DecCnn.fixed_unpooling, mcnet.py:240-256: x lands on even (2i, 2j), zeros elsewhere."""
    B, C, H, W = x.shape
    out = x.new_zeros(B, C, 2 * H, 2 * W)
    out[:, :, 0::2, 0::2] = x
    return out


def dec_cnn(sd, p, comb, res1, res2, res3):
    """DecCnn.forward, mcnet.py:227-238."""
    x = fixed_unpooling(comb) + res3
    x = F.relu(_convt(sd, p + 'dec3.0', x))
    x = F.relu(_convt(sd, p + 'dec3.2', x))
    x = F.relu(_convt(sd, p + 'dec3.4', x))
    x = fixed_unpooling(x) + res2
    x = F.relu(_convt(sd, p + 'dec2.0', x))
    x = F.relu(_convt(sd, p + 'dec2.2', x))
    x = fixed_unpooling(x) + res1
    x = F.relu(_convt(sd, p + 'dec1.0', x))
    return torch.tanh(_convt(sd, p + 'dec1.2', x))


def conv_lstm_cell(sd, p, inp, state, forget_bias=1.0):
    """ConvLstmCell.forward, mcnet.py:281-294; padding (feature_size-1)/2 = 1 under Py2 (:278)."""
    c, h = torch.chunk(state, 2, dim=1)
    z = _conv(sd, p + 'conv', torch.cat((inp, h), dim=1), 1)
    i, j, f, o = torch.chunk(z, 4, dim=1)
    new_c = c * torch.sigmoid(f + forget_bias) + torch.sigmoid(i) * torch.tanh(j)
    new_h = torch.tanh(new_c) * torch.sigmoid(o)
    return new_h, torch.cat((new_c, new_h), dim=1)


def mcnet_forward(sd, p, c_dim, K, T, diff_in, xt):
    """MCNet.forward, mcnet.py:391-453.  diff_in [B,K-1,1,H,W], xt [B,C,H,W]."""
    diffs = [diff_in[:, t] for t in range(diff_in.shape[1])]
    B, _, H, W = xt.shape
    gf4 = sd[p + 'conv_lstm_cell.conv.weight'].shape[0] // 4
    # get_initial_conv_lstm_state, mcnet.py:378-388 (Py2 integer division)
    state = xt.new_zeros(B, 2 * gf4, H // 8, W // 8)
    h_dyn = res_m = None
    for t in range(K - 1):
        enc_h, res_m = motion_enc(sd, p + 'motion_enc.', diffs[t])
        h_dyn, state = conv_lstm_cell(sd, p + 'conv_lstm_cell.', enc_h, state)
    pred, dyn, cont, res = [], [], [], []
    for t in range(T):
        if t > 0:
            enc_h, res_m = motion_enc(sd, p + 'motion_enc.', diffs[-1])
            h_dyn, state = conv_lstm_cell(sd, p + 'conv_lstm_cell.', enc_h, state)
        h_cont, res_c = content_enc(sd, p + 'content_enc.', xt)
        h_tpl = comb_layers(sd, p + 'comb_layers.', h_dyn, h_cont)
        dyn.append(h_dyn)
        cont.append(h_cont)
        r1 = residual(sd, p + 'residual1.', res_m[0], res_c[0])
        r2 = residual(sd, p + 'residual2.', res_m[1], res_c[1])
        r3 = residual(sd, p + 'residual3.', res_m[2], res_c[2])
        res.append([r1, r2, r3])
        x_hat = dec_cnn(sd, p + 'dec_cnn.', h_tpl, r1, r2, r3)
        if c_dim == 3:
            x_hat_gray = bgr2gray(inverse_transform(x_hat))
            xt_gray = bgr2gray(inverse_transform(xt))
        else:
            x_hat_gray = inverse_transform(x_hat)
            xt_gray = inverse_transform(xt)
        diffs.append(x_hat_gray - xt_gray)
        xt = x_hat
        pred.append(x_hat)
    return pred, dyn, cont, res


# ----------------------------------------------------------------------------- tai.py:244-348
def basic_conv_block(sd, p, x, layers=3):
    """create_basic_conv_block, tai.py:244-263: layers x (conv3x3 + ReLU) at indices 0,2,4."""
    for i in range(layers):
        x = F.relu(_conv(sd, p + '%d' % (2 * i), x, 1))
    return x


def kernel_generator_block(sd, p, x, layers=3):
    """create_1d_kernel_generator_block, tai.py:266-286: 3x(conv+ReLU), bilinear x2, conv (no act)."""
    for i in range(layers):
        x = F.relu(_conv(sd, p + '%d' % (2 * i), x, 1))
    return _conv(sd, p + '%d' % (2 * layers + 1), _up2(x), 1)


def upsample_block(sd, p, x):
    """moduleUpsample[i], tai.py:334-346: bilinear x2, conv3x3, ReLU."""
    return F.relu(_conv(sd, p + '1', _up2(x), 1))


class SepconvFunction(torch.autograd.Function):
    """SeparableConvolution (SeparableConvolution.py:6-92) on the C oracle: forward = updateOutput (.cu:19-47),
    backward = updateGradV / updateGradH / updateGradI (.cu:49-162), returning (gI, gV, gH, None) as :89 does."""

    @staticmethod
    def forward(ctx, inp_padded, vertical, horizontal, ks):
        ctx.save_for_backward(inp_padded, vertical, horizontal)
        ctx.ks = ks
        out = sepconv_oracle.forward(inp_padded.detach().numpy(), vertical.detach().numpy(), horizontal.detach().numpy(), ks)
        return torch.from_numpy(out).to(inp_padded.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        inp_padded, vertical, horizontal = ctx.saved_tensors
        gI, gV, gH = sepconv_oracle.backward(grad_output.contiguous().numpy(), inp_padded.detach().numpy(),
                                             vertical.detach().numpy(), horizontal.detach().numpy(), ctx.ks)
        return torch.from_numpy(gI), torch.from_numpy(gV), torch.from_numpy(gH), None


def sepconv(inp_padded, vertical, horizontal, ks, f64=False):
    """SeparableConvolution.forward, SeparableConvolution.py:11-52 via the C oracle (differentiable through
    SepconvFunction when any operand requires grad: the training leg, oracle/train_oracle.py)."""
    if torch.is_grad_enabled() and (inp_padded.requires_grad or vertical.requires_grad or horizontal.requires_grad):
        return SepconvFunction.apply(inp_padded.contiguous(), vertical.contiguous(), horizontal.contiguous(), ks)
    out = sepconv_oracle.forward(inp_padded.detach().cpu().numpy(), vertical.detach().cpu().numpy(),
                                 horizontal.detach().cpu().numpy(), ks, f64=f64)
    return torch.from_numpy(out).to(inp_padded.dtype)


def tai_kernelnet_forward(sd, p, num_block, ks, in1, in2, dyn1, dyn2, cont1, cont2, merged_res, ratio,
                          rc_loc=4, layers=3):
    """TAI.forward, tai.py:174-237."""
    join = torch.cat([dyn1, dyn2, cont1, cont2], 1)
    convs, pools = [], []
    for i in range(num_block - 3):
        src = join if i == 0 else pools[-1]
        convs.append(basic_conv_block(sd, p + 'moduleConv.%d.' % i, src, layers))
        pools.append(F.avg_pool2d(convs[-1], 2, 2))
    combine = []
    for i in range(num_block - 1):
        src = pools[-1] if i == 0 else combine[-1]
        d = basic_conv_block(sd, p + 'moduleDeconv.%d.' % i, src, layers)
        if i == rc_loc - 1:
            rc = d.new_full((d.shape[0], 1, d.shape[2], d.shape[3]), ratio)
            d = torch.cat([d, rc], dim=1)
        u = upsample_block(sd, p + 'moduleUpsample.%d.' % i, d)
        if i < num_block - 3:
            combine.append(u + convs[num_block - 3 - i - 1])
        else:
            combine.append(u + merged_res[num_block - i - 1])
    feat = combine[-1]
    pad = int(np.floor(ks / 2.0))
    v1 = kernel_generator_block(sd, p + 'moduleVertical1.', feat, layers)
    h1 = kernel_generator_block(sd, p + 'moduleHorizontal1.', feat, layers)
    v2 = kernel_generator_block(sd, p + 'moduleVertical2.', feat, layers)
    h2 = kernel_generator_block(sd, p + 'moduleHorizontal2.', feat, layers)
    dot1 = sepconv(F.pad(in1, [pad] * 4, mode='replicate'), v1, h1, ks)
    dot2 = sepconv(F.pad(in2, [pad] * 4, mode='replicate'), v2, h2, ks)
    return dot1, dot2, (v1, h1, v2, h2)


def tai_forward(sd, c_dim, num_block, ks, T, preceding, following, return_taps=False):
    """TAIFillInModel.forward, tai.py:52-120.  preceding [B,K,C,H,W], following [B,F,C,H,W]."""
    K, Fn = preceding.shape[1], following.shape[1]
    xt = preceding[:, -1]
    xt_F = following[:, 0]
    gp = bgr2gray_batched(inverse_transform(preceding)) if c_dim > 1 else inverse_transform(preceding)
    diff_in = gp[:, 1:] - gp[:, :-1]
    gf = bgr2gray_batched(inverse_transform(following)) if c_dim > 1 else inverse_transform(following)
    rev = torch.flip(gf, dims=[1])
    diff_in_F = rev[:, 1:] - rev[:, :-1]

    f_pred, f_dyn, f_cont, f_res = mcnet_forward(sd, 'generator.', c_dim, K, T, diff_in, xt)
    b_pred, b_dyn, b_cont, b_res = mcnet_forward(sd, 'generator.', c_dim, Fn, T, diff_in_F, xt_F)
    b_pred, b_dyn, b_cont, b_res = b_pred[::-1], b_dyn[::-1], b_cont[::-1], b_res[::-1]

    w = np.linspace(0, 1, num=T + 2).tolist()[1:-1]
    comb, out1, out2, taps = [], [], [], []
    for t in range(T):
        merged = [residual(sd, 'merge_residual1.', f_res[t][0], b_res[t][0]),
                  residual(sd, 'merge_residual2.', f_res[t][1], b_res[t][1]),
                  residual(sd, 'merge_residual3.', f_res[t][2], b_res[t][2])]
        d1, d2, tp = tai_kernelnet_forward(sd, 'kernelnet.', num_block, ks, f_pred[t], b_pred[t], f_dyn[t],
                                           b_dyn[t], f_cont[t], b_cont[t], merged, 1 - w[t])
        out1.append(d1)
        out2.append(d2)
        taps.append(tp)
        comb.append(0.5 * d1 + 0.5 * d2)
    out = {
        'pred': torch.stack(comb, dim=1),
        'pred_forward': torch.stack(f_pred, dim=1),
        'pred_backward': torch.stack(b_pred, dim=1),
        'interp_net_outputs_1': torch.stack(out1, dim=1),
        'interp_net_outputs_2': torch.stack(out2, dim=1),
    }
    if return_taps:
        out['_taps'] = taps
    return out


# ----------------------------------------------------------------------------- losses.py:17-44
def gdl(inp, target):
    """GDL.forward with reduce=True, losses.py:17-44."""
    B = inp.shape[0]
    H, W = inp.shape[-2:]
    a = inp.reshape(-1, H, W)
    b = target.reshape(-1, H, W)
    wl = ((a[:, :, :-1] - a[:, :, 1:]) - (b[:, :, :-1] - b[:, :, 1:])).abs()[:, 1:, :]
    hl = ((a[:, 1:, :] - a[:, :-1, :]) - (b[:, 1:, :] - b[:, :-1, :])).abs()[:, :, 1:]
    return (wl + hl).reshape(B, -1).mean()
