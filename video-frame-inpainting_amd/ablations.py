"""The ablation / baseline fill-in models that are built from the same blocks as bi-TAI:

  bi-TWI   ``TimeWeightedInterpolationFillInModel``        reference src/models/twi/twi.py:15-231
           bi-TAI without the time-ratio input; blend (1-w) Dot1 + w Dot2; submodules named ``mcnet`` / ``interp_net``
  bi-TWA   ``BidirectionalTimeWeightedAverageFillInModel`` reference src/models/bi_twa/bi_twa.py:9-66
  bi-SA    ``BidirectionalSimpleAverageFillInModel``       reference src/models/bi_sa/bi_sa.py:9-60
  TW_P_F   ``TimeWeightedPFFillInModel``                   reference src/models/tw_p_f/tw_p_f.py:5-33

Same forward contract and state-dict keys as the reference classes; they reuse the MI355X path of tai.py (fused
bidirectional MC-Net pass, HIP sepconv / upsample / bias+activation kernels)."""
import torch
import torch.nn as nn

from .mcnet import MCNet, Residual
from .tai import TAI, bidirectional_inputs, generate_both_directions, middle_frame_weights


class TimeWeightedInterpolationFillInModel(nn.Module):
    """twi.py:15-120: bidirectional prediction, time-AGNOSTIC kernel network, time-weighted blend."""

    def __init__(self, gf_dim, c_dim, feature_size, ks, num_block=5, kf_dim=32, layers=3, forget_bias=1, bias=True):
        super().__init__()
        self.conv_lstm_state_size = 8 * gf_dim
        self.mcnet = MCNet(gf_dim, c_dim, feature_size, forget_bias=forget_bias, bias=bias)
        self.merge_residual3 = Residual(gf_dim * 8, kf_dim * 4)
        self.merge_residual2 = Residual(gf_dim * 4, kf_dim * 2)
        self.merge_residual1 = Residual(gf_dim * 2, kf_dim * 1)
        self.interp_net = TAI(gf_dim, ks, num_block, layers, kf_dim, rc_loc=-1)
        self.fuse_directions = True

    def forward(self, T, preceding_frames, following_frames):
        K, Fn = preceding_frames.size(1), following_frames.size(1)
        diff_in, xt, diff_in_F, xt_F = bidirectional_inputs(preceding_frames, following_frames)
        (f_pred, f_dyn, f_cont, f_res), (b_pred, b_dyn, b_cont, b_res) = generate_both_directions(
            self.mcnet, K, Fn, T, diff_in, xt, diff_in_F, xt_F, fuse=self.fuse_directions)
        w = middle_frame_weights(T)
        combination, out1, out2 = [], [], []
        for t in range(T):
            merged = {1: self.merge_residual2(f_res[t][1], b_res[t][1]), 2: self.merge_residual3(f_res[t][2], b_res[t][2])}
            dot1, dot2 = self.interp_net(f_pred[t].contiguous(), b_pred[t].contiguous(), f_dyn[t], b_dyn[t], f_cont[t],
                                         b_cont[t], merged)
            out1.append(dot1)
            out2.append(dot2)
            combination.append((1 - w[t]) * dot1 + w[t] * dot2)          # twi.py:105
        return {
            'pred': torch.stack(combination, dim=1),
            'pred_forward': torch.stack(f_pred, dim=1),
            'pred_backward': torch.stack(b_pred, dim=1),
            'interp_net_outputs_1': torch.stack(out1, dim=1),
            'interp_net_outputs_2': torch.stack(out2, dim=1),
        }


class _BidirectionalAverage(nn.Module):
    def __init__(self, gf_dim, c_dim, feature_size, forget_bias=1, bias=True):
        super().__init__()
        self.c_dim = c_dim
        self.conv_lstm_state_size = 8 * gf_dim
        self.generator = MCNet(gf_dim, c_dim, feature_size, forget_bias=forget_bias, bias=bias)
        self.fuse_directions = True

    def _weights(self, T):
        raise NotImplementedError

    def forward(self, T, preceding_frames, following_frames):
        K, Fn = preceding_frames.size(1), following_frames.size(1)
        diff_in, xt, diff_in_F, xt_F = bidirectional_inputs(preceding_frames, following_frames)
        (f_pred, _, _, _), (b_pred, _, _, _) = generate_both_directions(
            self.generator, K, Fn, T, diff_in, xt, diff_in_F, xt_F, fuse=self.fuse_directions)
        w = self._weights(T)
        combination = [(1 - w[t]) * f_pred[t] + w[t] * b_pred[t] for t in range(T)]
        return {'pred': torch.stack(combination, dim=1), 'pred_forward': torch.stack(f_pred, dim=1),
                'pred_backward': torch.stack(b_pred, dim=1)}


class BidirectionalTimeWeightedAverageFillInModel(_BidirectionalAverage):
    """bi_twa.py:9-66: (1 - w[t]) forward + w[t] backward."""

    def _weights(self, T):
        return middle_frame_weights(T)


class BidirectionalSimpleAverageFillInModel(_BidirectionalAverage):
    """bi_sa.py:9-60: 0.5 forward + 0.5 backward."""

    def _weights(self, T):
        return [0.5] * T


class TimeWeightedPFFillInModel(nn.Module):
    """tw_p_f.py:5-33: middle frame t = (1 - w[t]) last preceding frame + w[t] first following frame; no parameters."""

    def forward(self, T, preceding_frames, following_frames):
        last_p, first_f = preceding_frames[:, -1:], following_frames[:, :1]
        w = middle_frame_weights(T)
        return {'pred': torch.cat([(1 - w[t]) * last_p + w[t] * first_f for t in range(T)], dim=1)}
