"""Colour / range helpers and weight init used on the hot path (reference src/util/util.py:22-41,188-196)."""
import numpy as np
import torch
import torch.nn as nn
from torch.nn import init

# BGR -> gray weights, channel order as stored: B, G, R (reference util.py:32,39; they sum to 0.9999)
_GRAY_BGR = (0.1140, 0.5870, 0.2989)


def inverse_transform(images):
    """[-1, 1] -> [0, 1] (util.py:22-23)."""
    return (images + 1.) / 2


def fore_transform(images):
    """[0, 1] -> [-1, 1] (util.py:26-27)."""
    return images * 2 - 1


def bgr2gray(image):
    """[B,3,H,W] BGR -> [B,1,H,W] (util.py:30-34)."""
    b, g, r = _GRAY_BGR
    return (b * image[:, 0] + g * image[:, 1] + r * image[:, 2]).unsqueeze(1)


def bgr2gray_batched(image):
    """[B,T,3,H,W] BGR -> [B,T,1,H,W] (util.py:37-41)."""
    b, g, r = _GRAY_BGR
    return (b * image[:, :, 0] + g * image[:, :, 1] + r * image[:, :, 2]).unsqueeze(2)


def gray01(frames):
    """Frames in [-1,1] ([..., C, H, W], C in {1,3}) -> gray in [0,1] with a singleton channel: the
    composition the reference applies before every temporal difference (tai.py:67-71, mcnet.py:439-444)."""
    x = inverse_transform(frames)
    if frames.shape[-3] == 1:
        return x
    b, g, r = _GRAY_BGR
    return (b * x[..., 0, :, :] + g * x[..., 1, :, :] + r * x[..., 2, :, :]).unsqueeze(-3)


def weights_init(m):
    """Xavier-normal (gain 1) conv / conv-transpose weights with zero bias; U(0, 0.02) linear weights
    (util.py:193-199).  Spectral-norm layers are handled by their own classes' ``reset``."""
    from .sn_discriminator import SNConv2d, SNLinear
    # written through the parameters themselves (not ``.data``) under no_grad, so their version counters move and
    # conv_ops rebuilds the derived (Winograd-domain / flipped) weights it caches per version
    with torch.no_grad():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, SNConv2d)):
            init.xavier_normal_(m.weight, gain=1)
            if m.bias is not None:
                init.constant_(m.bias, 0.0)
        elif isinstance(m, (nn.Linear, SNLinear)):
            init.uniform_(m.weight, 0.0, 0.02)
            init.constant_(m.bias, 0.0)


def move_to_devices(model, device=None):
    """The reference's ``model.cuda()`` (util.py:188-190); ``device`` lets one process per GPU pick its card."""
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return model.to(device)


def to_numpy(tensor, transpose=None):
    """util.py:172-185."""
    arr = tensor.detach().cpu().numpy()
    if transpose is not None:
        arr = np.transpose(arr, transpose)
    return arr


def frames_to_uint8(video):
    """[T,C,H,W] in [-1,1] -> uint8 [T,H,W,C] with the reference's TRUNCATING cast
    (predict.py:113-118; train.py:279-280)."""
    clipped = np.clip(to_numpy(video, transpose=(0, 2, 3, 1)), -1, 1)
    return (255 * inverse_transform(clipped)).astype(np.uint8)
