"""Winograd F(4x4, 3x3) on the fp32 MFMA pipe (csrc/wino43_conv.hip.inc; the wide layers of the inference path, conv_ops.set_winograd_tile) against an fp64
convolution of the same operands, against the F(2x2, 3x3) kernel, and -- the whole bi-TAI forward -- against the CPU oracle.
Reference layers: nn.Conv2d(C, K, 3, padding=1) (+ReLU) of src/models/mcnet/mcnet.py:79-118,131-152,165-176,198-224 and
src/models/tai/tai.py:248-286."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

_ACT = {None: 0, 'relu': 1, 'tanh': 2}
# F(4x4, 3x3) in fp32 on the points (0, +-3/4, +-3/2, inf): against the magnitude sum of a dot product its error is ~3-4x F(2x2, 3x3)'s
# (bound there: 4e-6, tests/test_gpu_wino_conv.py; with Lavin's points 0, +-1, +-2, inf it was ~15x and this bound 6e-5)
TOL = 2e-5


def _run(x_parts, w, b, act):
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, cp, H, W = x_parts[0].shape
    C, K = cp * len(x_parts), w.shape[0]
    s = torch.cuda.current_stream().cuda_stream
    U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino43_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'transform')
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    if len(x_parts) == 1:
        _native.check(L.tai_conv3x3_wino43_forward(x_parts[0].data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, _ACT[act], s), 'forward')
    else:
        ptrs = (ctypes.c_void_p * len(x_parts))(*[p.data_ptr() for p in x_parts])
        _native.check(L.tai_conv3x3_wino43_forward_parts(ptrs, len(x_parts), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, _ACT[act], s),
                      'forward_parts')
    return y


def _operands(N, C, K, H, W, seed=0):
    g = torch.Generator().manual_seed(seed + N + C + K + H)
    x = torch.randn(N, C, H, W, generator=g).cuda()
    w = (torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    return x, w, b


# (N, C, K, H, W): one tile, tiles that straddle images and workgroups, K that is not a multiple of 64, C = 4, bi-TAI shapes
# ... and channel counts that are not multiples of 4 (the kernel network's 51 -> 51 and 65 -> 64 layers: zero-padded weights)
SHAPES = [(1, 4, 64, 4, 4), (2, 8, 70, 12, 12), (3, 128, 128, 8, 20), (5, 12, 3, 4, 8), (2, 256, 256, 32, 32), (8, 512, 130, 16, 16),
          (4, 128, 256, 64, 64), (3, 51, 51, 16, 24), (2, 65, 64, 8, 12), (2, 1, 5, 4, 4)]


@pytest.mark.parametrize('act', [None, 'relu', 'tanh'])
@pytest.mark.parametrize('shape', SHAPES)
def test_wino43_matches_fp64_conv(shape, act):
    x, w, b = _operands(*shape)
    got = _run([x], w, b, act)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = torch.relu(ref) if act == 'relu' else (torch.tanh(ref) if act == 'tanh' else ref)
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=1)
    assert torch.isfinite(got).all()
    err = float(((got.double() - ref).abs() / (1 + mag)).max())
    assert err <= TOL, err


def test_wino43_sees_every_tap_and_the_padding():
    """One tap at a time on small-integer operands (G = 1/6, 1/24: unlike F(2x2, 3x3) the transformed weights are not exact in binary,
    so the comparison has a bound: 1e-5 of the largest output), every pixel including the zero-padded border."""
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-2, 3, (2, 8, 8, 12), generator=g).float().cuda()
    b = torch.randint(-3, 4, (64,), generator=g).float().cuda()
    for ty in range(3):
        for tx in range(3):
            w = torch.zeros(64, 8, 3, 3)
            w[:, :, ty, tx] = torch.randint(-1, 2, (64, 8), generator=g).float()
            w = w.cuda()
            want = F.conv2d(x, w, b, padding=1)
            assert float((_run([x], w, b, None) - want).abs().max()) <= 1e-5 * float(want.abs().max()), (ty, tx)


@pytest.mark.parametrize('nparts', [2, 4])
def test_wino43_reads_channel_parts_without_a_cat(nparts):
    N, cp, K, H, W = 3, 64, 128, 16, 24
    g = torch.Generator().manual_seed(nparts)
    parts = [torch.randn(N, cp, H, W, generator=g).cuda() for _ in range(nparts)]
    w = (torch.randn(K, cp * nparts, 3, 3, generator=g) * 0.05).cuda()
    b = torch.randn(K, generator=g).cuda()
    assert torch.equal(_run(parts, w, b, 'relu'), _run([torch.cat(parts, 1)], w, b, 'relu'))


@pytest.mark.parametrize('shape,nparts', [((3, 128, 128, 16, 24), 1), ((2, 64, 70, 8, 8), 2), ((4, 256, 128, 32, 32), 2)])
def test_wino43_pooled_and_unpool_add_outputs(shape, nparts):
    """The second outputs of the general entry point against the plain launch: ypool = max_pool2d(relu(y), 2) exactly; y2 = y +
    fixed_unpooling(addx) exactly (the plain output untouched); the sum alone in y when y2 is NULL."""
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    N, C, K, H, W = shape
    x, w, b = _operands(*shape)
    parts = [c.contiguous() for c in x.chunk(nparts, 1)]
    s = torch.cuda.current_stream().cuda_stream
    U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, C), device='cuda')
    _native.check(L.tai_conv3x3_wino43_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'transform')
    ptrs = (ctypes.c_void_p * nparts)(*[p.data_ptr() for p in parts])
    for act in (None, 'relu'):
        plain = _run(parts, w, b, act)
        y = torch.full((N, K, H, W), float('nan'), device='cuda')
        yp = torch.full((N, K, H // 2, W // 2), float('nan'), device='cuda')
        _native.check(L.tai_conv3x3_wino43_forward_ex(ptrs, nparts, U.data_ptr(), b.data_ptr(), y.data_ptr(), yp.data_ptr(), None, None, N, C, K, H, W,
                                                      _ACT[act], s), 'ex pool')
        assert torch.equal(y, plain) and torch.equal(yp, F.max_pool2d(plain, 2))
    plain = _run(parts, w, b, None)
    addx = torch.randn(N, K, H // 2, W // 2, generator=torch.Generator().manual_seed(5)).cuda()
    want = plain.clone()
    want[:, :, 0::2, 0::2] += addx
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    y2 = torch.full((N, K, H, W), float('nan'), device='cuda')
    _native.check(L.tai_conv3x3_wino43_forward_ex(ptrs, nparts, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, addx.data_ptr(), y2.data_ptr(), N, C, K, H,
                                                  W, 0, s), 'ex add')
    assert torch.equal(y, plain) and torch.equal(y2, want)
    y.fill_(float('nan'))
    _native.check(L.tai_conv3x3_wino43_forward_ex(ptrs, nparts, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, addx.data_ptr(), None, N, C, K, H, W, 0, s),
                  'ex sum')
    assert torch.equal(y, want)
    # refusals: y2 without addx, addx with an activation
    assert L.tai_conv3x3_wino43_forward_ex(ptrs, nparts, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, y2.data_ptr(), N, C, K, H, W, 0, s) != 0
    assert L.tai_conv3x3_wino43_forward_ex(ptrs, nparts, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, addx.data_ptr(), None, N, C, K, H, W, 1, s) != 0


def test_conv_ops_second_output_layers_take_the_4x4_tile(monkeypatch):
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO43_MIN_WORKGROUPS', 1)
    x, w, b = _operands(16, 128, 128, 32, 32)
    addx = torch.randn(16, 128, 16, 16, generator=torch.Generator().manual_seed(9)).cuda()
    with torch.no_grad():
        before = conv_ops.set_winograd_tile(2)
        y2, yp2 = conv_ops.conv_bias_act_maxpool(x, w, b, 1, 'relu')
        p2, s2 = conv_ops.conv_bias_unpool_add(x, w, b, 1, addx)
        conv_ops.set_winograd_tile(4)
        try:
            y4, yp4 = conv_ops.conv_bias_act_maxpool(x, w, b, 1, 'relu')
            p4, s4 = conv_ops.conv_bias_unpool_add((x[:, :64].contiguous(), x[:, 64:].contiguous()), w, b, 1, addx)
            _, s4only = conv_ops.conv_bias_unpool_add(x, w, b, 1, addx, keep_plain=False)
        finally:
            conv_ops.set_winograd_tile(before)
    assert torch.equal(y4, _run([x], w, b, 'relu')) and torch.equal(yp4, F.max_pool2d(y4, 2)) and not torch.equal(y4, y2)
    assert torch.equal(p4, _run([x], w, b, None)) and torch.equal(s4only, s4) and not torch.equal(p4, p2)
    for a, c in ((y4, y2), (yp4, yp2), (p4, p2), (s4, s2)):
        assert float((a - c).abs().max()) <= 1e-4 * float(c.abs().max())


def test_wino43_rejects_what_it_cannot_run():
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    x, w, b = _operands(1, 8, 64, 8, 8)
    y = torch.empty(1, 64, 8, 8, device='cuda')
    U = torch.empty(L.tai_conv3x3_wino43_weight_floats(64, 8), device='cuda')
    s = torch.cuda.current_stream().cuda_stream
    assert L.tai_conv3x3_wino43_weight_floats(64, 6) == 36 * 64 * 8                                           # C padded to a multiple of 4
    assert L.tai_conv3x3_wino43_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 8, 64, 6, 8, 0, s) != 0      # H % 4
    xs2 = (ctypes.c_void_p * 2)(x.data_ptr(), x.data_ptr())
    assert L.tai_conv3x3_wino43_forward_parts(xs2, 2, U.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 12, 64, 8, 8, 0, s) != 0     # parts of 6 channels
    assert L.tai_conv3x3_wino43_forward(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 8, 64, 8, 8, 3, s) != 0      # act
    assert L.tai_conv3x3_wino43_forward(None, U.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 8, 64, 8, 8, 0, s) != 0


def test_wino43_is_bit_reproducible():
    from video_frame_inpainting_amd import _native
    L = _native.lib()
    x, w, b = _operands(8, 256, 256, 32, 32)
    first = _run([x], w, b, 'relu').clone()
    for _ in range(20):
        assert torch.equal(_run([x], w, b, 'relu'), first)
    assert L.tai_conv3x3_wino43_set_waves(3) == -1 and L.tai_conv3x3_wino43_set_waves(8) == -1      # round 4's forms are gone
    assert L.tai_conv3x3_wino43_set_waves(0) == 0


def test_conv_ops_takes_the_4x4_tile_only_when_asked_and_only_on_wide_layers(monkeypatch):
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO43_MIN_WORKGROUPS', 1)
    x, w, b = _operands(16, 128, 128, 32, 32)          # (conv_ops leaves layers of fewer than 72 F(2x2, 3x3) workgroups to MIOpen)
    xs, ws, bs = _operands(32, 48, 48, 32, 32)
    with torch.no_grad():
        before = conv_ops.set_winograd_tile(2)
        y2, y2s = conv_ops.conv_bias_act(x, w, b, 1, 'relu'), conv_ops.conv_bias_act(xs, ws, bs, 1, 'relu')
        prev = conv_ops.set_winograd_tile(4)
        try:
            assert before == 4 and prev == 2 and conv_ops.get_winograd_tile() == 4
            y4, y4s = conv_ops.conv_bias_act(x, w, b, 1, 'relu'), conv_ops.conv_bias_act(xs, ws, bs, 1, 'relu')
            y4p = conv_ops.conv_bias_act((x[:, :64].contiguous(), x[:, 64:].contiguous()), w, b, 1, 'relu')
            # a narrow layer outside MC-Net's recurrence (the kernel network, the merge residuals: conv_ops.mark_outside_recurrence)
            narrow = torch.nn.Conv2d(48, 48, 3, padding=1).cuda()
            with torch.no_grad():
                narrow.weight.copy_(ws)
                narrow.bias.copy_(bs)
            conv_ops.mark_outside_recurrence(narrow)
            y4m = conv_ops.conv_bias_act(xs, narrow.weight, narrow.bias, 1, 'relu')
        finally:
            conv_ops.set_winograd_tile(before)
    assert torch.equal(y4s, y2s)                                  # C = K = 48 < WINO43_MIN_CHANNELS: stays on F(2x2, 3x3) ...
    assert not torch.equal(y4m, y2s) and torch.equal(y4m, _run([xs], ws, bs, 'relu'))       # ... unless the layer is marked
    assert not torch.equal(y4, y2) and torch.equal(y4, _run([x], w, b, 'relu')) and torch.equal(y4p, y4)
    assert float((y4 - y2).abs().max()) <= 1e-4 * float(y2.abs().max())
    with pytest.raises(ValueError):
        conv_ops.set_winograd_tile(3)


def test_full_width_forward_with_the_4x4_tile_matches_cpu_oracle(monkeypatch):
    """The whole bi-TAI forward (TAI_gray, full width, 128x128, K = F = T = 5) with F(4x4, 3x3) on every layer it takes, against the CPU
    oracle with the bound of the default path's own test (tests/test_gpu_model.py: 1e-4 of each output's magnitude; PSNR within 0.01 dB,
    SSIM within 1e-4), and next to the default forward of the same weights."""
    import numpy as np
    import video_frame_inpainting_amd as vfi
    from video_frame_inpainting_amd import conv_ops, metrics, synthetic
    from oracle import tai_oracle
    monkeypatch.setattr(conv_ops, 'WINO43_MIN_WORKGROUPS', 1)     # two clips: few workgroups per layer
    m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    clips = synthetic.make_clips(2, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
    P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, 5, 5))
    with torch.no_grad():
        ref = tai_oracle.tai_forward(sd, 1, 5, 51, 5, P, Fo)
        m.to('cuda:0').eval()
        before = conv_ops.set_winograd_tile(2)
        try:
            o2 = m(5, P.cuda(), Fo.cuda())
            conv_ops.set_winograd_tile(4)
            o4 = m(5, P.cuda(), Fo.cuda())
        finally:
            conv_ops.set_winograd_tile(before)
    for k in ('pred', 'pred_forward', 'pred_backward', 'interp_net_outputs_1', 'interp_net_outputs_2'):
        scale = float(ref[k].abs().max())
        e4, e2 = float((o4[k].cpu() - ref[k]).abs().max()), float((o2[k].cpu() - ref[k]).abs().max())
        print('%s: max |gpu - oracle| / max |oracle|: tile 4 %.2e, tile 2 %.2e' % (k, e4 / scale, e2 / scale))
        assert e4 <= 1e-4 * scale, (k, e4, e2, scale)
        assert not torch.equal(o4[k], o2[k])                      # the F(4x4, 3x3) kernel did run
    p4, s4, _ = metrics.compute_errors(o4['pred'].cpu().numpy(), GT.numpy())
    pr, sr, _ = metrics.compute_errors(ref['pred'].numpy(), GT.numpy())
    assert np.abs(np.asarray(p4) - np.asarray(pr)).max() <= 0.01 and np.abs(np.asarray(s4) - np.asarray(sr)).max() <= 1e-4


@pytest.mark.parametrize('k,N,Cin,K,H,W', [(5, 3, 8, 70, 8, 12), (7, 2, 16, 64, 16, 16), (5, 5, 64, 128, 64, 64), (7, 3, 128, 256, 32, 32)])
def test_wino43_displaced_read_blocks_match_fp64_kxk_conv(k, N, Cin, K, H, W):
    """MotionEnc's 5 x 5 / 7 x 7 "same" convolutions (mcnet.py:36-47) on the 4 x 4 tile: the halo-carrying plane read S x S times
    (tai_conv3x3_wino43_forward_blocks) against an fp64 k x k convolution; the pooled output plain and into the window of a larger
    plane whose halo must stay untouched."""
    from video_frame_inpainting_amd import _native, conv_ops
    L = _native.lib()
    g = torch.Generator().manual_seed(k + N + Cin)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    w = (torch.randn(K, Cin, k, k, generator=g) * (2.0 / (k * k * Cin)) ** 0.5).cuda()
    b = torch.randn(K, generator=g).cuda()
    S, top, left, in_h, in_w = conv_ops.halo_geometry(H, W, k)
    plane = torch.zeros(N, Cin, in_h, in_w, device='cuda')
    plane[:, :, top:top + H, left:left + W] = x
    wb = conv_ops._block3x3_weight(w)
    s = torch.cuda.current_stream().cuda_stream
    U = torch.empty(L.tai_conv3x3_wino43_weight_floats(K, S * S * Cin), device='cuda')
    _native.check(L.tai_conv3x3_wino43_transform_weights(wb.data_ptr(), U.data_ptr(), K, S * S * Cin, s), 'transform')
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=k // 2))
    mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=k // 2)
    y = torch.full((N, K, H, W), float('nan'), device='cuda')
    yp = torch.full((N, K, H // 2, W // 2), float('nan'), device='cuda')
    _native.check(L.tai_conv3x3_wino43_forward_blocks(plane.data_ptr(), k, U.data_ptr(), b.data_ptr(), y.data_ptr(), yp.data_ptr(), 0, 0, 0, 0,
                                                      N, S * S * Cin, K, H, W, in_h, in_w, 1, 2, 1, s), 'blocks')
    assert torch.isfinite(y).all()
    assert float(((y.double() - ref).abs() / (1 + mag)).max()) <= TOL
    assert torch.equal(yp, F.max_pool2d(y, 2))
    # ... the pooled output into the window (3, 4) of a plane with a sentinel halo, no plain output requested twice
    ph, pw = H // 2 + 7, W // 2 + 10
    big = torch.full((N, K, ph, pw), 7.0, device='cuda')
    y2 = torch.empty_like(y)
    _native.check(L.tai_conv3x3_wino43_forward_blocks(plane.data_ptr(), k, U.data_ptr(), b.data_ptr(), y2.data_ptr(), big.data_ptr(), ph, pw, 3, 4,
                                                      N, S * S * Cin, K, H, W, in_h, in_w, 1, 2, 1, s), 'blocks window')
    assert torch.equal(y2, y) and torch.equal(big[:, :, 3:3 + H // 2, 4:4 + W // 2], yp)
    big[:, :, 3:3 + H // 2, 4:4 + W // 2] = 7.0
    assert bool((big == 7.0).all())
    # refusals: a plane without the halo, an odd window
    assert L.tai_conv3x3_wino43_forward_blocks(plane.data_ptr(), k, U.data_ptr(), b.data_ptr(), y.data_ptr(), None, 0, 0, 0, 0,
                                               N, S * S * Cin, K, H, W, in_h - 1, in_w, 1, 2, 1, s) != 0
    assert L.tai_conv3x3_wino43_forward_blocks(plane.data_ptr(), k, U.data_ptr(), b.data_ptr(), y.data_ptr(), big.data_ptr(), ph, pw, 3, 3,
                                               N, S * S * Cin, K, H, W, in_h, in_w, 1, 2, 1, s) != 0


def test_motion_enc_chain_on_the_4x4_tile_matches_the_2x2_tile(monkeypatch):
    """conv_ops.motion_enc_chain (MotionEnc's three stages, each pooled output written into the next stage's halo plane) with the 5x5 / 7x7
    stages on tai_conv3x3_wino43_forward_blocks against the same chain on F(2x2, 3x3)."""
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO43_MIN_WORKGROUPS', 1)
    monkeypatch.setattr(conv_ops, 'WINO_MIN_WORKGROUPS', 1)
    torch.manual_seed(4)
    c1, c2, c3 = (torch.nn.Conv2d(1, 16, 5, padding=2).cuda(), torch.nn.Conv2d(16, 32, 5, padding=2).cuda(),
                  torch.nn.Conv2d(32, 64, 7, padding=3).cuda())
    diff = torch.randn(6, 1, 64, 64, device='cuda')
    outs = {}
    with torch.no_grad():
        for tile in (2, 4):
            prev = conv_ops.set_winograd_tile(tile)
            try:
                p3, res = conv_ops.motion_enc_chain(diff, c1, c2, c3)
                outs[tile] = [p3.clone()] + [r.clone() for r in res]
            finally:
                conv_ops.set_winograd_tile(prev)
        want = F.max_pool2d(torch.relu(c3(F.max_pool2d(torch.relu(c2(F.max_pool2d(torch.relu(c1(diff)), 2))), 2))), 2)
    assert not torch.equal(outs[4][0], outs[2][0])
    for a, b in zip(outs[4], outs[2]):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max())
    assert float((outs[4][0] - want).abs().max()) <= 1e-4 * float(want.abs().max())


def test_workgroup_placement_changes_nothing_but_the_order():
    """tai_conv3x3_wino43_set_placement: the XCD-aware mapping of workgroups to (tile block, channel block) / (split, c block, k block) is a
    permutation of the plain dispatch order -- forward, displaced-read blocks and the weight gradient give the same bits under both, for
    grids that are and are not multiples of 8 and for 1, 2, 3, 4, 8 and 16 channel blocks."""
    from video_frame_inpainting_amd import _native, conv_ops
    L = _native.lib()
    assert L.tai_conv3x3_wino43_set_placement(1) == 1                      # the default
    try:
        for (N, C, K, H, W) in ((8, 64, 64, 32, 32), (8, 64, 128, 32, 32), (3, 32, 130, 16, 16), (8, 32, 256, 16, 16), (4, 16, 512, 16, 16),
                                (2, 16, 1024, 16, 16), (5, 20, 64, 12, 20)):
            x, w, b = _operands(N, C, K, H, W)
            go = torch.randn(N, K, H, W, device='cuda')
            res = {}
            for v in (1, 0):
                L.tai_conv3x3_wino43_set_placement(v)
                res[v] = (_run([x], w, b, 'relu').clone(), conv_ops.wino_weight_grad(x, go) if W % 16 == 0 else None)
            assert torch.equal(res[0][0], res[1][0]), (N, C, K, H, W)
            assert res[0][1] is None or torch.equal(res[0][1], res[1][1]), (N, C, K, H, W)
    finally:
        L.tai_conv3x3_wino43_set_placement(1)
