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
