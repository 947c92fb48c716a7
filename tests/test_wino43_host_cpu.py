"""Host side of the opt-in F(4x4, 3x3) path (conv_ops.set_winograd_tile): which layers take it.  No GPU, no compute calls."""
import pytest


def test_tile_switch_and_layer_selection():
    from video_frame_inpainting_amd import conv_ops
    assert conv_ops.get_winograd_tile() == 4                              # the default: F(4x4, 3x3) on the wide layers
    with pytest.raises(ValueError):
        conv_ops.set_winograd_tile(3)
    assert conv_ops.set_winograd_tile(2) == 4
    assert not conv_ops._wino43_ok(64, 256, 256, 32, 32)                 # tile 2: every layer on F(2x2, 3x3)
    prev = conv_ops.set_winograd_tile(4)
    try:
        assert prev == 2 and conv_ops.get_winograd_tile() == 4
        assert conv_ops._wino43_ok(64, 256, 256, 32, 32)                 # 128 tile blocks x 4 channel blocks = 512 workgroups
        assert conv_ops._wino43_ok(64, 512, 1024, 16, 16, nparts=2)      # the ConvLSTM's (input, h)
        assert conv_ops._wino43_ok(64, 64, 64, 128, 128)                 # MC-Net's first block (WINO43_MIN_CHANNELS = 64 since the points change)
        assert not conv_ops._wino43_ok(64, 32, 256, 32, 32)              # C < 64
        assert not conv_ops._wino43_ok(64, 256, 32, 32, 32)              # K < 64
        assert not conv_ops._wino43_ok(64, 256, 256, 30, 32)             # H % 4
        assert not conv_ops._wino43_ok(64, 512, 256, 16, 16, nparts=2)   # 32 x 4 = 128 workgroups: too few (the threshold is 150)
        assert conv_ops._wino43_ok(160, 256, 256, 16, 16)                # 80 x 4 = 320 (round 4's threshold of 400 refused it)
        # a layer outside MC-Net's recurrence (kernel network, merge residuals) takes the 4 x 4 tile at any width, any channel count
        import torch
        narrow = torch.nn.Conv2d(51, 51, 3, padding=1)
        assert not conv_ops._wino43_ok(160, 51, 51, 128, 128, 1, narrow.weight)
        conv_ops.mark_outside_recurrence(narrow)
        assert conv_ops._wino43_ok(160, 51, 51, 128, 128, 1, narrow.weight)
        assert not conv_ops._wino43_ok(160, 8, 51, 128, 128, 1, narrow.weight)          # fewer than 16 input channels: not worth a chunk loop
        assert not conv_ops._wino43_ok(160, 102, 51, 128, 128, 2, narrow.weight)        # parts whose channel count is no multiple of 4
        assert conv_ops._wino43_blocks_ok(64, 128, 256, 32, 32) and not conv_ops._wino43_blocks_ok(2, 128, 256, 32, 32)
        assert not conv_ops._wino43_ok(64, 384, 256, 32, 32, nparts=4) or (384 // 4) % 4 == 0
        assert not conv_ops._wino43_ok(4096, 1024, 1024, 32, 32)         # 2^32 elements: beyond the kernel's 32-bit offsets
    finally:
        conv_ops.set_winograd_tile(4)
    assert conv_ops.get_winograd_tile() == 4


def test_predict_option_exists():
    from video_frame_inpainting_amd.options import TestOptions
    opt = TestOptions().parse(['--name', 'x', '--model_key', 'TAI_gray', '--K', '5', '--T', '5', '--F', '5', '--qual_result_root', '/tmp/q', '--winograd_tile', '4',
                              '--synthetic', '1'], allow_unknown=True, require_gpu=False)
    assert opt.winograd_tile == 4


@pytest.mark.parametrize('C, K, H, W', [(3, 5, 16, 16), (4, 2, 8, 12), (1, 3, 4, 4)])
def test_discriminator_layer_as_a_3x3_layer_on_space_to_depth_planes(C, K, H, W):
    """The host side of the discriminator's in-tree route (sn_discriminator._s2d_weight / _s2d_weight_grad; SNDiscriminator.py:113-133): a
    4x4, stride-2, padding-1 layer equals the 3x3, stride-1, padding-1 layer over F.pixel_unshuffle(x, 2) with the rearranged weight -- output,
    input gradient (pixel_shuffle of the 3x3 layer's) and weight gradient (folded back) -- in float64, exactly up to summation order."""
    import torch
    import torch.nn.functional as F
    from video_frame_inpainting_amd.sn_discriminator import _s2d_applies, _s2d_weight, _s2d_weight_grad
    g = torch.Generator().manual_seed(C + K + H)
    x = torch.randn(2, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(K, C, 4, 4, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, w, None, 2, 1)
    w3 = _s2d_weight(w)
    assert w3.shape == (K, 4 * C, 3, 3) and int((w3 != 0).sum()) == K * C * 16          # 20 of the 36 slots per channel are structural zeros
    assert float((y - F.conv2d(F.pixel_unshuffle(x, 2), w3, None, 1, 1)).abs().max()) < 1e-12
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    gw, = torch.autograd.grad(y, w, gy)
    w3d = w3.detach().requires_grad_()
    g3, = torch.autograd.grad(F.conv2d(F.pixel_unshuffle(x, 2), w3d, None, 1, 1), w3d, gy)
    assert float((gw - _s2d_weight_grad(g3, C)).abs().max()) < 1e-12
    xr = x.clone().requires_grad_()
    gx, = torch.autograd.grad(F.conv2d(xr, w.detach(), None, 2, 1), xr, gy)
    gxs = F.conv2d(gy, w3.detach().transpose(0, 1).flip(2, 3), None, 1, 1)
    assert float((gx - F.pixel_shuffle(gxs, 2)).abs().max()) < 1e-12
    assert not _s2d_applies(x, w, (2, 2), (1, 1))                                        # host tensors keep the reference's route
