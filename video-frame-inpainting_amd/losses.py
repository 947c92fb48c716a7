"""Image gradient difference loss (reference src/losses/losses.py:4-44, Mathieu et al.)."""
import torch
import torch.nn as nn


class GDL(nn.Module):
    """L1 distance between the horizontal and vertical finite differences of prediction and target, each cropped to the
    common (H-1) x (W-1) window, summed; mean over everything when ``reduce`` (losses.py:30-43)."""

    def __init__(self, reduce=True):
        super().__init__()
        self.reduce = reduce

    def forward(self, input, target):
        B = input.size(0)
        H, W = input.shape[-2:]
        lead = input.shape[:-2]
        a = input.reshape(-1, H, W)
        b = target.reshape(-1, H, W)
        # d/dx (sign as in the reference: left minus right), rows 1..H-1;  d/dy (lower minus upper), cols 1..W-1
        dw = ((a[:, 1:, :-1] - a[:, 1:, 1:]) - (b[:, 1:, :-1] - b[:, 1:, 1:])).abs()
        dh = ((a[:, 1:, 1:] - a[:, :-1, 1:]) - (b[:, 1:, 1:] - b[:, :-1, 1:])).abs()
        loss = (dw + dh).reshape(*lead, H - 1, W - 1)
        return loss.reshape(B, -1).mean() if self.reduce else loss
