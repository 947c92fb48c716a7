"""MC-Net generator blocks (Villegas et al.) as used by bi-TAI: the MI355X-side mirror of the reference's
``src/models/mcnet/mcnet.py``.

State-dict keys are the reference's (so ``snapshot['generator']`` loads unchanged, environments.py:113):
``motion_enc.dyn_conv{1.0,2.1,3.1}``, ``conv_lstm_cell.conv``, ``content_enc.cont_conv{1.0,1.2,2.1,2.3,3.1,3.3,3.5}``,
``comb_layers.h_comb.{0,2,4}``, ``residual{1,2,3}.res.{0,2}``, ``dec_cnn.dec3.{0,2,4}``, ``dec_cnn.dec2.{0,2}``,
``dec_cnn.dec1.{0,2}``.  The convolutions ride MIOpen through ``torch.nn.functional``; everything around them is
arranged for a static, graph-capturable schedule:
  * layers are kept in ``IndexedConvs`` containers that own only the parameterised layers (at the reference's
    Sequential indices) and run conv+activation chains without Python-side module dispatch per activation;
  * the reference's ``fixed_unpooling`` (zero-stuffed 2x upsample via permute/cat/view, mcnet.py:240-256) followed by
    ``+ res`` is one strided in-place add into a copy of the residual;
  * ``ConvTranspose2d(k=3, s=1, p=1)`` runs as the equivalent direct convolution with the transposed, flipped kernel
    (MIOpen's forward-conv path), keeping the reference's [in, out, 3, 3] parameter layout;
  * the ConvLSTM state is allocated on the input's device with integer H//8 x W//8 (the reference relies on Py2
    integer division, mcnet.py:278,384).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .conv_ops import conv_bias_act, conv_bias_act_maxpool, conv_bias_unpool_add, motion_enc_chain
from .util import gray01


class IndexedConvs(nn.Module):
    """Conv layers registered under explicit integer names, e.g. {0, 2, 4} or {1}, matching the indices the
    reference's nn.Sequential gave them (activations and pools own no parameters and are not modules here)."""

    def __init__(self, layers):
        super().__init__()
        self.order = []
        for index, layer in layers:
            self.add_module(str(index), layer)
            self.order.append(str(index))

    def convs(self):
        return [getattr(self, k) for k in self.order]


def _conv_relu_chain(x, convs, last_act='relu'):
    """conv + bias + ReLU for every layer but the last, which gets ``last_act`` ('relu', 'tanh' or None).  ``x`` may be a
    tuple of tensors standing for their concatenation along the channels (conv_ops.conv_bias_act)."""
    for i, c in enumerate(convs):
        act = 'relu' if i + 1 < len(convs) else last_act
        x = conv_bias_act(x, c.weight, c.bias, c.padding[0], act)
    return x


def _conv_relu_chain_pooled(x, convs):
    """_conv_relu_chain with ReLU on every layer, returning (y, max_pool2d(y, 2)); the pool is fused into the last
    convolution where the kernel allows it."""
    for c in convs[:-1]:
        x = conv_bias_act(x, c.weight, c.bias, c.padding[0], 'relu')
    c = convs[-1]
    return conv_bias_act_maxpool(x, c.weight, c.bias, c.padding[0], 'relu')


def _convt3x3_as_conv(x, layer, act):
    # ConvTranspose2d(cin, cout, 3, stride 1, padding 1)  ==  conv2d with weight[o, i, ky, kx] = wt[i, o, 2-ky, 2-kx]
    return conv_bias_act(x, layer.weight, layer.bias, 1, act, transposed=True)


def unpool2x_add(x, res):
    """fixed_unpooling(x) + res  (mcnet.py:234-236, 240-256): x lands on the even (2i, 2j) sites."""
    if (x.is_cuda and x.dtype == torch.float32 and res.dtype == torch.float32 and x.shape[3] % 2 == 0
            and not (torch.is_grad_enabled() and (x.requires_grad or res.requires_grad))):
        from . import _native
        x, res = x.contiguous(), res.contiguous()
        out = torch.empty_like(res)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().tai_unpool2x_add(x.data_ptr(), res.data_ptr(), out.data_ptr(), x.shape[0] * x.shape[1],
                                                         x.shape[2], x.shape[3], torch.cuda.current_stream(x.device).cuda_stream),
                          'tai_unpool2x_add')
        return out
    if (x.is_cuda and x.dtype == torch.float32 and res.dtype == torch.float32 and x.shape[3] % 2 == 0
            and tuple(res.shape) == (x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3])):
        return _Unpool2xAdd.apply(x, res)        # training: the same kernel under autograd
    out = res.clone()
    out[:, :, 0::2, 0::2] += x
    return out


class _Unpool2xAdd(torch.autograd.Function):
    """``fixed_unpooling(x) + res`` under autograd on ``tai_unpool2x_add``.  As ATen ops (clone, an in-place add on a strided slice)
    the training path copied ``res`` once forwards and went through CopySlices backwards; here the forward is the one pass of the
    inference path and the backward is the identity for ``res`` and a stride-2 gather for ``x``."""

    @staticmethod
    def forward(ctx, x, res):
        from . import _native
        x, res = x.contiguous(), res.contiguous()
        out = torch.empty_like(res)
        with torch.cuda.device(x.device):
            _native.check(_native.lib().tai_unpool2x_add(x.data_ptr(), res.data_ptr(), out.data_ptr(), x.shape[0] * x.shape[1],
                                                         x.shape[2], x.shape[3], torch.cuda.current_stream(x.device).cuda_stream),
                          'tai_unpool2x_add')
        return out

    @staticmethod
    def backward(ctx, g):
        gx = g[:, :, 0::2, 0::2].contiguous() if ctx.needs_input_grad[0] else None
        return gx, (g if ctx.needs_input_grad[1] else None)


class MotionEnc(nn.Module):
    """mcnet.py:14-60: conv5x5(1->gf)+ReLU; pool, conv5x5(gf->2gf)+ReLU; pool, conv7x7(2gf->4gf)+ReLU; pool."""

    def __init__(self, gf_dim):
        super().__init__()
        self.dyn_conv1 = IndexedConvs([(0, nn.Conv2d(1, gf_dim, 5, padding=2))])
        self.dyn_conv2 = IndexedConvs([(1, nn.Conv2d(gf_dim, gf_dim * 2, 5, padding=2))])
        self.dyn_conv3 = IndexedConvs([(1, nn.Conv2d(gf_dim * 2, gf_dim * 4, 7, padding=3))])

    def forward(self, input_diff):
        fused = motion_enc_chain(input_diff, self.dyn_conv1.convs()[0], self.dyn_conv2.convs()[0], self.dyn_conv3.convs()[0])
        if fused is not None:       # the three stages hand their pooled outputs over in halo-carrying planes (conv_ops)
            return fused
        c1, p1 = _conv_relu_chain_pooled(input_diff, self.dyn_conv1.convs())
        c2, p2 = _conv_relu_chain_pooled(p1, self.dyn_conv2.convs())
        c3, p3 = _conv_relu_chain_pooled(p2, self.dyn_conv3.convs())
        return p3, [c1, c2, c3]


class ContentEnc(nn.Module):
    """mcnet.py:63-119: VGG-like 3x3 stacks C->gf->gf /2 ->2gf->2gf /2 ->4gf->4gf->4gf /2."""

    def __init__(self, c_dim, gf_dim):
        super().__init__()
        g = gf_dim
        self.cont_conv1 = IndexedConvs([(0, nn.Conv2d(c_dim, g, 3, padding=1)), (2, nn.Conv2d(g, g, 3, padding=1))])
        self.cont_conv2 = IndexedConvs([(1, nn.Conv2d(g, g * 2, 3, padding=1)), (3, nn.Conv2d(g * 2, g * 2, 3, padding=1))])
        self.cont_conv3 = IndexedConvs([(1, nn.Conv2d(g * 2, g * 4, 3, padding=1)), (3, nn.Conv2d(g * 4, g * 4, 3, padding=1)),
                                        (5, nn.Conv2d(g * 4, g * 4, 3, padding=1))])

    def forward(self, raw):
        c1, p1 = _conv_relu_chain_pooled(raw, self.cont_conv1.convs())
        c2, p2 = _conv_relu_chain_pooled(p1, self.cont_conv2.convs())
        c3, p3 = _conv_relu_chain_pooled(p2, self.cont_conv3.convs())
        return p3, [c1, c2, c3]


class CombLayers(nn.Module):
    """mcnet.py:122-153: cat(h_dyn, h_cont) -> 8gf->4gf->2gf->4gf, 3x3 + ReLU each."""

    def __init__(self, gf_dim):
        super().__init__()
        g = gf_dim
        self.h_comb = IndexedConvs([(0, nn.Conv2d(g * 8, g * 4, 3, padding=1)), (2, nn.Conv2d(g * 4, g * 2, 3, padding=1)),
                                    (4, nn.Conv2d(g * 2, g * 4, 3, padding=1))])

    def forward(self, h_dyn, h_cont):
        return _conv_relu_chain((h_dyn, h_cont), self.h_comb.convs())          # conv(cat(..)) without the cat


class Residual(nn.Module):
    """mcnet.py:156-185: cat -> conv3x3 + ReLU -> conv3x3 (no final activation)."""

    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.res = IndexedConvs([(0, nn.Conv2d(in_dim, out_dim, 3, padding=1)), (2, nn.Conv2d(out_dim, out_dim, 3, padding=1))])

    def forward(self, input_dyn, input_cont):
        return _conv_relu_chain((input_dyn, input_cont), self.res.convs(), last_act=None)   # conv(cat(..)) without the cat

    def forward_into(self, input_dyn, input_cont, out):
        """forward() with the result written into ``out`` (a batch slice of a larger buffer): no cat afterwards."""
        c0, c1 = self.res.convs()
        mid = conv_bias_act((input_dyn, input_cont), c0.weight, c0.bias, c0.padding[0], 'relu')
        return conv_bias_act(mid, c1.weight, c1.bias, c1.padding[0], None, out=out)

    def forward_unpool_add(self, input_dyn, input_cont, below, keep_res=True):
        """(res, res + fixed_unpooling(below)): the block's output and the sum DecCnn forms from it (mcnet.py:234-236);
        ``keep_res=False`` returns (None, sum)."""
        c0, c1 = self.res.convs()
        mid = conv_bias_act((input_dyn, input_cont), c0.weight, c0.bias, c0.padding[0], 'relu')
        return conv_bias_unpool_add(mid, c1.weight, c1.bias, c1.padding[0], below, keep_plain=keep_res)


class DecCnn(nn.Module):
    """mcnet.py:188-256: three unpool(+residual) stages of 3x3 transposed convs, ReLU inside, Tanh at the end."""

    def __init__(self, c_dim, gf_dim):
        super().__init__()
        g = gf_dim
        T = lambda i, o: nn.ConvTranspose2d(i, o, 3, padding=1)
        self.dec3 = IndexedConvs([(0, T(g * 4, g * 4)), (2, T(g * 4, g * 4)), (4, T(g * 4, g * 2))])
        self.dec2 = IndexedConvs([(0, T(g * 2, g * 2)), (2, T(g * 2, g))])
        self.dec1 = IndexedConvs([(0, T(g, g)), (2, T(g, c_dim))])

    @staticmethod
    def _stage(x, layers, last_act):
        for i, layer in enumerate(layers):
            x = _convt3x3_as_conv(x, layer, 'tanh' if (last_act == 'tanh' and i + 1 == len(layers)) else 'relu')
        return x

    def forward(self, comb, res1, res2, res3):
        x = self._stage(unpool2x_add(comb, res3), self.dec3.convs(), 'relu')
        x = self._stage(unpool2x_add(x, res2), self.dec2.convs(), 'relu')
        return self._stage(unpool2x_add(x, res1), self.dec1.convs(), 'tanh')

    def stage(self, index, x):
        """The transposed-convolution stack of stage 3, 2 or 1 on an input that already is unpool(below) + residual."""
        return self._stage(x, (self.dec1, self.dec2, self.dec3)[index - 1].convs(), 'tanh' if index == 1 else 'relu')

    fixed_unpooling = staticmethod(lambda x: unpool2x_add(x, x.new_zeros(x.shape[0], x.shape[1], 2 * x.shape[2], 2 * x.shape[3])))


class _LstmGates(torch.autograd.Function):
    """(new_c, new_h) from the gates tensor and c under autograd: tai_convlstm_gates_forward / _backward, one kernel each."""

    @staticmethod
    def forward(ctx, gates, c, forget_bias):
        from . import _native
        gates, c = gates.contiguous(), c.contiguous()
        N, F4, H, W = gates.shape
        new_c, new_h = torch.empty_like(c), torch.empty_like(c)
        with torch.cuda.device(gates.device):
            _native.check(_native.lib().tai_convlstm_gates_forward(
                gates.data_ptr(), c.data_ptr(), new_c.data_ptr(), new_h.data_ptr(), N, F4 // 4, H * W, float(forget_bias),
                torch.cuda.current_stream(gates.device).cuda_stream), 'tai_convlstm_gates_forward')
        ctx.forget_bias = float(forget_bias)
        ctx.save_for_backward(gates, c, new_c)
        return new_c, new_h

    @staticmethod
    def backward(ctx, g_c, g_h):
        from . import _native
        gates, c, new_c = ctx.saved_tensors
        N, F4, H, W = gates.shape
        if g_c is None and g_h is None:
            return None, None, None
        g_c = g_c.contiguous() if g_c is not None else None
        g_h = g_h.contiguous() if g_h is not None else None
        d_gates, d_c = torch.empty_like(gates), torch.empty_like(c)
        with torch.cuda.device(gates.device):
            _native.check(_native.lib().tai_convlstm_gates_backward(
                gates.data_ptr(), c.data_ptr(), new_c.data_ptr(), g_c.data_ptr() if g_c is not None else None,
                g_h.data_ptr() if g_h is not None else None, d_gates.data_ptr(), d_c.data_ptr(), N, F4 // 4, H * W, ctx.forget_bias,
                torch.cuda.current_stream(gates.device).cuda_stream), 'tai_convlstm_gates_backward')
        return d_gates, d_c, None


class ConvLstmCell(nn.Module):
    """mcnet.py:259-294.  state = cat(c, h); gates (i, j, f, o) = chunks of conv(cat(input, h));
    c' = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j);  h' = tanh(c') * sigmoid(o)."""

    def __init__(self, feature_size, num_features, forget_bias=1, bias=True):
        super().__init__()
        self.feature_size = feature_size
        self.num_features = num_features
        self.forget_bias = forget_bias
        self.conv = nn.Conv2d(num_features * 2, num_features * 4, feature_size, padding=(feature_size - 1) // 2, bias=bias)

    def forward(self, input, state, state_is_zero=False):
        """``state`` is the reference's cat(c, h) tensor or a (c, h) pair; returns (new_h, state) with ``state`` in the
        form it came in.  MCNet.forward keeps the pair: no cat / chunk copies, and conv(cat(input, h)) reads its two
        operands in place.  ``state_is_zero``: the caller knows h == 0 (the first step of a sequence, mcnet.py:378-388), so
        the h half of the reduction -- half the convolution -- contributes exact zeros and is left out."""
        as_pair = isinstance(state, (tuple, list))
        c, h = state if as_pair else torch.chunk(state, 2, dim=1)
        if state_is_zero and not (torch.is_grad_enabled() and self.conv.weight.requires_grad):
            from .conv_ops import _cached
            F_ = self.num_features
            w_in = _cached(self.conv.weight, ('input_half', F_), lambda: self.conv.weight.detach()[:, :F_].contiguous())
            gates = conv_bias_act(input, w_in, self.conv.bias, self.conv.padding[0], None)
            return self._gates(gates, c, as_pair)
        x = (input, h) if (as_pair or h.is_contiguous()) else torch.cat((input, h), dim=1)
        gates = conv_bias_act(x, self.conv.weight, self.conv.bias, self.conv.padding[0], None)
        return self._gates(gates, c, as_pair)

    def _gates(self, gates, c, as_pair):
        N, F4, H, W = gates.shape
        fused = (gates.is_cuda and gates.dtype == torch.float32 and (H * W) % 4 == 0 and c.dtype == torch.float32
                 and not (torch.is_grad_enabled() and (gates.requires_grad or c.requires_grad)))
        if fused:
            from . import _native
            gates, c = gates.contiguous(), c.contiguous()
            new_c, new_h = torch.empty_like(c), torch.empty_like(c)
            with torch.cuda.device(gates.device):
                _native.check(_native.lib().tai_convlstm_gates_forward(
                    gates.data_ptr(), c.data_ptr(), new_c.data_ptr(), new_h.data_ptr(), N, F4 // 4, H * W,
                    float(self.forget_bias), torch.cuda.current_stream(gates.device).cuda_stream), 'tai_convlstm_gates_forward')
        elif gates.is_cuda and gates.dtype == torch.float32 and (H * W) % 4 == 0 and c.dtype == torch.float32:
            new_c, new_h = _LstmGates.apply(gates, c, self.forget_bias)           # training: the same kernel, and its gradient
        else:
            i, j, f, o = torch.chunk(gates, 4, dim=1)
            new_c = c * torch.sigmoid(f + self.forget_bias) + torch.sigmoid(i) * torch.tanh(j)
            new_h = torch.tanh(new_c) * torch.sigmoid(o)
        return new_h, ((new_c, new_h) if as_pair else torch.cat((new_c, new_h), dim=1))


class MCNet(nn.Module):
    """mcnet.py:350-453: K-1 motion-encoder/ConvLSTM steps over the given difference frames, then T autoregressive
    steps (content encoder, combination layers, residuals, decoder), feeding gray(x_hat) - gray(x_t) back in."""

    def __init__(self, gf_dim, c_dim, feature_size, forget_bias=1, bias=True):
        super().__init__()
        self.c_dim = c_dim
        self.gf_dim = gf_dim
        self.motion_enc = MotionEnc(gf_dim)
        self.conv_lstm_cell = ConvLstmCell(feature_size, 4 * gf_dim, forget_bias=forget_bias, bias=bias)
        self.content_enc = ContentEnc(c_dim, gf_dim)
        self.comb_layers = CombLayers(gf_dim)
        self.residual3 = Residual(gf_dim * 8, gf_dim * 4)
        self.residual2 = Residual(gf_dim * 4, gf_dim * 2)
        self.residual1 = Residual(gf_dim * 2, gf_dim * 1)
        self.dec_cnn = DecCnn(c_dim, gf_dim)
        # res[t][0], the full-resolution residual, is part of what the reference's MCNet.forward returns (mcnet.py:453) but
        # no model of the path reads it (tai.py:224-226 only reaches indices 2 and 1); a fill-in model that knows this sets
        # keep_res1 = False and gets None in that slot: the residual then exists only inside the decoder's sum
        self.keep_res1 = True

    def get_initial_conv_lstm_state(self, batch_size, image_size, like):
        return like.new_zeros(batch_size, 8 * self.gf_dim, image_size[0] // 8, image_size[1] // 8)

    def forward(self, K, T, diff_in, xt):
        """diff_in [B, K-1, 1, H, W] gray differences; xt [B, C, H, W] last known frame.
        Returns lists pred[T], dyn[T], cont[T], res[T] = [res1, res2, res3]."""
        assert K >= 2, 'MC-Net needs at least two input frames (one difference frame)'
        diffs = [diff_in[:, t] for t in range(diff_in.shape[1])]
        state = torch.chunk(self.get_initial_conv_lstm_state(xt.shape[0], xt.shape[2:4], xt), 2, dim=1)
        state = (state[0].contiguous(), state[1].contiguous())      # (c, h) kept apart: see ConvLstmCell.forward
        h_dyn = res_m = None
        for t in range(K - 1):
            enc_h, res_m = self.motion_enc(diffs[t])
            h_dyn, state = self.conv_lstm_cell(enc_h, state, state_is_zero=(t == 0))
        pred, dyn, cont, res = [], [], [], []
        xt_gray = gray01(xt)
        for t in range(T):
            if t > 0:
                enc_h, res_m = self.motion_enc(diffs[-1])
                h_dyn, state = self.conv_lstm_cell(enc_h, state)
            h_cont, res_c = self.content_enc(xt)
            h_tpl = self.comb_layers(h_dyn, h_cont)
            dyn.append(h_dyn)
            cont.append(h_cont)
            # Residual blocks and decoder stages interleaved (the reference computes the three residuals first, then
            # dec_cnn(h_tpl, res_1, res_2, res_3): mcnet.py:431-436): each residual's last convolution also emits
            # unpool(stage below) + residual, the input of its decoder stage
            res_3, u3 = self.residual3.forward_unpool_add(res_m[2], res_c[2], h_tpl)
            res_2, u2 = self.residual2.forward_unpool_add(res_m[1], res_c[1], self.dec_cnn.stage(3, u3))
            res_1, u1 = self.residual1.forward_unpool_add(res_m[0], res_c[0], self.dec_cnn.stage(2, u2), keep_res=self.keep_res1)
            res.append([res_1, res_2, res_3])
            x_hat = self.dec_cnn.stage(1, u1)
            x_hat_gray = gray01(x_hat)
            diffs.append(x_hat_gray - xt_gray)
            xt, xt_gray = x_hat, x_hat_gray
            pred.append(x_hat)
        return pred, dyn, cont, res


class MCNetFillInModel(nn.Module):
    """mcnet.py:301-347: forward-only MC-Net baseline behind the fill-in model interface."""

    def __init__(self, gf_dim, c_dim, feature_size, forget_bias=1, bias=True):
        super().__init__()
        self.c_dim = c_dim
        self.generator = MCNet(gf_dim, c_dim, feature_size, forget_bias=forget_bias, bias=bias)

    def forward(self, T, preceding_frames, following_frames):
        K = preceding_frames.size(1)
        gray = gray01(preceding_frames)
        pred, _, _, _ = self.generator(K, T, gray[:, 1:] - gray[:, :-1], preceding_frames[:, -1])
        return {'pred': torch.stack(pred, dim=1)}
