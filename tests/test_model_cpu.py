"""Host-side logic of the product package on CPU: block numerics against the reference's golden vectors, state-dict
schema, model registry, flags, losses, metrics, the C-ABI library's symbol table.  (The sepconv op itself has no CPU
form -- like the reference -- so whole-model runs here plug the ORACLE's sepconv into the product model: that pins the
product's control flow around the op; the op's own parity is in the -m gpu tests.)"""
import ctypes
import os

import numpy as np
import pytest
import torch

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import _native, losses, mcnet, metrics, options, parallel, synthetic, tai, util
from video_frame_inpainting_amd.sn_discriminator import SNConv2d, SNDiscriminator, max_singular_value

from oracle import sepconv_oracle

TOL = dict(rtol=1e-5, atol=2e-6)


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def _block(z, name):
    pre = name + '/'
    d = {k[len(pre):]: torch.from_numpy(v) for k, v in z.items() if k.startswith(pre)}
    return d, {k[2:]: v for k, v in d.items() if k.startswith('w/')}


@pytest.fixture(scope='module')
def blocks(golden_dir):
    return _load(golden_dir, 'blocks.npz')


def _close(a, b, **kw):
    np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), **(kw or TOL))


def test_motion_and_content_enc(blocks):
    d, sd = _block(blocks, 'motion_enc')
    m = mcnet.MotionEnc(8); m.load_state_dict(sd)
    out, res = m(d['x'])
    _close(out, d['out'])
    for i in range(3):
        _close(res[i], d['res%d' % i])
    d, sd = _block(blocks, 'content_enc')
    m = mcnet.ContentEnc(3, 8); m.load_state_dict(sd)
    out, res = m(d['x'])
    _close(out, d['out'])
    for i in range(3):
        _close(res[i], d['res%d' % i])


def test_comb_residual_dec_lstm(blocks):
    d, sd = _block(blocks, 'comb_layers')
    m = mcnet.CombLayers(8); m.load_state_dict(sd)
    _close(m(d['a'], d['b']), d['out'])
    d, sd = _block(blocks, 'residual')
    m = mcnet.Residual(32, 16); m.load_state_dict(sd)
    _close(m(d['a'], d['b']), d['out'])
    d, sd = _block(blocks, 'dec_cnn')
    m = mcnet.DecCnn(3, 8); m.load_state_dict(sd)
    _close(m(d['comb'], d['r1'], d['r2'], d['r3']), d['out'])
    _close(mcnet.DecCnn.fixed_unpooling(d['comb']), d['unpool'])
    d, sd = _block(blocks, 'conv_lstm')
    m = mcnet.ConvLstmCell(3, 32); m.load_state_dict(sd)
    h, ns = m(d['inp'], d['state'])
    _close(h, d['h'])
    _close(ns, d['new_state'])


def test_kernel_network_blocks(blocks):
    d, sd = _block(blocks, 'basic_conv_block')
    m = tai.create_basic_conv_block(3, 12, 8); m.load_state_dict(sd)
    _close(mcnet._conv_relu_chain(d['x'], m.convs()), d['out'])
    d, sd = _block(blocks, 'kernel_generator_block')
    m = tai.create_1d_kernel_generator_block(3, 4, 51); m.load_state_dict(sd)
    _close(m(d['x']), d['out'])
    _, ups = tai.create_decoder_blocks(4, 4, 3, 4)
    for i in (0, 3):
        d, sd = _block(blocks, 'upsample_block_%d' % i)
        ups[i].load_state_dict(sd)
        _close(ups[i](d['x']), d['out'])


def test_util_and_gdl(blocks):
    d, _ = _block(blocks, 'util')
    _close(util.inverse_transform(d['x5']), d['inv'])
    _close(util.bgr2gray_batched(d['x5']), d['gray_b'])
    _close(util.bgr2gray(d['x5'][:, 0]), d['gray'])
    _close(util.gray01(d['x5']), util.bgr2gray_batched(util.inverse_transform(d['x5'])))
    d, _ = _block(blocks, 'gdl')
    _close(losses.GDL()(d['a'], d['b']).reshape(1), d['out'])


def _oracle_sepconv(inp, v, h, ks=51):
    return torch.from_numpy(sepconv_oracle.forward(inp.numpy(), v.numpy(), h.numpy(), ks))


def test_mcnet_forward_matches_reference_run(golden_dir):
    z = _load(golden_dir, 'mcnet_gray.npz')
    sd = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('w/')}
    m = mcnet.MCNetFillInModel(4, 1, 3); m.load_state_dict(sd)
    P = torch.from_numpy(z['P'])
    with torch.no_grad():
        pred, dyn, cont, res = m.generator(4, 3, (P[:, 1:] - P[:, :-1]) / 2, P[:, -1])
    np.testing.assert_allclose(torch.stack(pred, 1).numpy(), z['pred'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.stack(dyn, 1).numpy(), z['dyn'], rtol=1e-4, atol=1e-5)
    for t in range(3):
        for i in range(3):
            np.testing.assert_allclose(res[t][i].numpy(), z['res_%d_%d' % (t, i)], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('tag,fuse', [('gray', True), ('gray', False), ('color', True)])
def test_tai_forward_control_flow_matches_reference_run(golden_dir, tag, fuse):
    z = _load(golden_dir, 'tai_%s.npz' % tag)
    sd = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('w/')}
    m = vfi.TAIFillInModel(4, int(z['c_dim'][0]), 3, int(z['ks'][0]), num_block=int(z['num_block'][0]), kf_dim=2)
    m.load_state_dict(sd)                       # the reference's key schema loads unchanged (strict)
    m.fuse_directions = fuse
    m.batch_time_steps = fuse            # (False, False) is the reference's literal per-direction, per-step schedule
    m.kernelnet.separableConvolution = _oracle_sepconv
    with torch.no_grad():
        out = m(int(z['T'][0]), torch.from_numpy(z['P']), torch.from_numpy(z['F']))
    for k in ('pred', 'pred_forward', 'pred_backward', 'interp_net_outputs_1', 'interp_net_outputs_2'):
        np.testing.assert_allclose(out[k].numpy(), z['out/' + k], rtol=1e-4, atol=2e-5, err_msg=k)


def test_state_dict_schema_of_named_models():
    g = vfi.create_model('TAI_gray').state_dict()
    c = vfi.create_model('TAI_color').state_dict()
    assert len(g) == 142 and sum(v.numel() for v in g.values()) == 38320861
    assert len(c) == 128 and sum(v.numel() for v in c.values()) == 22393759
    assert tuple(g['kernelnet.moduleConv.0.0.weight'].shape) == (256, 1024, 3, 3)
    assert tuple(g['kernelnet.moduleUpsample.3.1.weight'].shape) == (64, 65, 3, 3)
    assert tuple(g['generator.conv_lstm_cell.conv.weight'].shape) == (1024, 512, 3, 3)
    assert tuple(g['generator.dec_cnn.dec3.4.weight'].shape) == (256, 128, 3, 3)     # ConvTranspose: [in, out]
    assert 'merge_residual1.res.0.weight' in g
    assert 'kernelnet.moduleVertical1.7.bias' in g and 'kernelnet.moduleHorizontal2.4.weight' in g


def test_create_model_keys_and_json(tmp_path):
    assert isinstance(vfi.create_model('MCNet_gray'), vfi.MCNetFillInModel)
    spec = '{"class": "TAIFillInModel", "args": [4, 1, 3, 7], "kwargs": {"num_block": 4, "kf_dim": 2}}'
    assert isinstance(vfi.create_model(spec), vfi.TAIFillInModel)
    p = tmp_path / 'm.json'
    p.write_text(spec)
    assert isinstance(vfi.create_model(str(p)), vfi.TAIFillInModel)
    with pytest.raises(RuntimeError):
        vfi.create_model('not a model')
    with pytest.raises(RuntimeError):
        vfi.create_model('{"class": "SloMoFillInModel", "args": [32, 3], "kwargs": {}}')


def test_sepconv_op_has_no_cpu_path_and_keeps_the_reference_asserts():
    inp, v, h = torch.zeros(1, 1, 58, 58), torch.zeros(1, 51, 8, 8), torch.zeros(1, 51, 8, 8)
    with pytest.raises(NotImplementedError):               # SeparableConvolution.py:48-49
        vfi.SeparableConvolution.apply(inp, v, h, 51)
    with pytest.raises(AssertionError):                    # :27 height relation
        vfi.SeparableConvolution.apply(torch.zeros(1, 1, 57, 58), v, h, 51)
    with pytest.raises(AssertionError):                    # :29 filter size
        vfi.SeparableConvolution.apply(inp, v[:, :50], h, 51)
    with pytest.raises(AssertionError):                    # :31 contiguity
        vfi.SeparableConvolution.apply(inp.transpose(2, 3), v, h, 51)


def test_capi_library_exports_every_declared_symbol():
    syms = _native.declared_symbols()
    assert {'tai_sepconv_forward', 'tai_sepconv_backward', 'tai_sepconv_last_error', 'tai_sepconv_version',
            'tai_sepconv_forward_bytes', 'tai_sepconv_backward_bytes', 'tai_sepconv_set_forward_variant'} <= set(syms)
    assert os.path.exists(_native.LIB_PATH), 'run __graft_entry__.build() first'
    L = ctypes.CDLL(_native.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), s
    L.tai_sepconv_forward_bytes.restype = ctypes.c_longlong
    L.tai_sepconv_backward_bytes.restype = ctypes.c_longlong
    assert L.tai_sepconv_forward_bytes(1, 1, 128, 128, 51) == 6876944          # SURVEY.md 8d, per sample
    assert L.tai_sepconv_forward_bytes(1, 3, 256, 256, 51) == 28648752
    assert L.tai_sepconv_backward_bytes(1, 1, 128, 128, 51) == 13688352
    assert L.tai_sepconv_version() >= 100


def test_options_surface():
    o = options.TestOptions().parse(['--K', '5', '--T', '5', '--F', '5', '--model_key', 'TAI_gray', '--qual_result_root',
                                     'r', '--image_size', '128', '--c_dim', '1'], require_gpu=False)
    assert o.image_size == [128, 128] and o.padding_size == [0, 0] and o.snapshot_file_name == 'model_best.ckpt'
    assert o.batch_size == 4 and o.checkpoints_dir == 'checkpoints' and not o.intermediate_preds
    t = options.TrainOptions().parse(['--K', '5', '--T', '5', '--F', '5', '--model_key', 'TAI_gray', '--image_size', '64',
                                      '80'], require_gpu=False)
    assert (t.lr, t.beta1, t.alpha, t.beta, t.df_dim, t.Ip, t.disc_window_size) == (1e-4, 0.5, 1.0, 0.02, 64, 3, 3)
    assert t.image_size == [64, 80]
    with pytest.raises(AssertionError):
        options.TestOptions().parse(['--K', '5', '--T', '5', '--F', '5', '--model_key', 'x', '--qual_result_root', 'r'])


def test_sn_discriminator_semantics():
    torch.manual_seed(0)
    W = torch.randn(6, 20)
    sigma, u = max_singular_value(W, None, Ip=50)
    assert abs(float(sigma) - float(torch.linalg.svdvals(W)[0])) < 1e-3
    conv = SNConv2d(3, 4, 4, stride=2, padding=1, Ip=3)
    w0 = conv.weight.detach().clone()
    x = torch.randn(2, 3, 8, 8)
    conv(x)
    w1 = conv.weight.detach().clone()
    assert conv.u is not None and not torch.allclose(w0, w1)          # weight overwritten in place ...
    conv(x)
    assert not torch.allclose(w1, conv.weight.detach())               # ... cumulatively, on every call
    assert 'u' not in conv.state_dict()
    d = SNDiscriminator((32, 32), 1, 3, 4, 3)
    assert sorted(k for k in d.state_dict()) == sorted(
        ['conv_layers.%d.%s' % (i, p) for i in (0, 2, 4, 6) for p in ('weight', 'bias')] +
        ['linear_layer.weight', 'linear_layer.bias'])
    out = d(torch.randn(2, 7, 1, 32, 32))
    assert out.shape == (2, 5)
    out.sum().backward()
    assert all(p.grad is not None for p in d.parameters())


def test_fake_labels_and_loss_composition():
    from video_frame_inpainting_amd.environments import L2GDLDiscTrainingEnvironment as E
    e = E.__new__(E)
    e.K, e.T, e.F, e.disc_t = 5, 5, 5, 3
    assert e.create_fake_labels().tolist() == [1, 1, 1] + [0] * 7 + [1, 1, 1]     # environments.py:308-323
    e.K, e.T, e.F = 2, 3, 2
    assert e.create_fake_labels().tolist() == [0] * 5


def test_metrics_definition():
    rs = np.random.RandomState(0)
    a = rs.randint(0, 256, (40, 36)).astype(np.uint8)
    b = np.clip(a.astype(int) + rs.randint(-9, 10, a.shape), 0, 255).astype(np.uint8)
    mse = np.mean((a.astype(float) - b.astype(float)) ** 2)
    assert abs(metrics.psnr_uint8(a, b) - 10 * np.log10(255 ** 2 / mse)) < 1e-12
    from scipy.ndimage import uniform_filter        # independent restatement of skimage 0.13.1 compare_ssim defaults

    def ssim_ref(X, Y, win=7):
        X, Y = X.astype(np.float64), Y.astype(np.float64)
        n = win * win
        cn = n / (n - 1.0)
        ux, uy = uniform_filter(X, win), uniform_filter(Y, win)
        vx = cn * (uniform_filter(X * X, win) - ux * ux)
        vy = cn * (uniform_filter(Y * Y, win) - uy * uy)
        vxy = cn * (uniform_filter(X * Y, win) - ux * uy)
        C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        p = (win - 1) // 2
        return S[p:-p, p:-p].mean()
    assert abs(metrics.ssim_uint8(a, b) - ssim_ref(a, b)) < 1e-10
    assert metrics.ssim_uint8(a, a) == pytest.approx(1.0)
    assert metrics.to_uint8(np.array([-1.0, 0.0, 0.999, 1.0, 3.0])).tolist() == [0, 127, 254, 255, 255]   # truncation
    x = rs.uniform(-1, 1, (2, 3, 1, 16, 16)).astype(np.float32)
    p, s, l2 = metrics.compute_errors(x, x)
    assert np.all(np.isinf(p)) and np.allclose(s, 1) and np.allclose(l2, 0)


def test_synthetic_clips_are_deterministic_and_in_range():
    a = synthetic.make_clips(2, 6, 3, 32, 32, 1002)
    b = synthetic.make_clips(2, 6, 3, 32, 32, 1002)
    assert a.shape == (2, 6, 3, 32, 32) and a.dtype == np.float32 and np.array_equal(a, b)
    assert a.min() >= -1 and a.max() <= 1 and not np.array_equal(a[:, 0], a[:, 1])
    P, M, F = synthetic.split_clip(a, 2, 2, 2)
    assert P.shape[1] == M.shape[1] == F.shape[1] == 2


def test_shard_slice_partitions_the_clips():
    for n in (0, 1, 7, 32, 33):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                got += list(range(n))[parallel.shard_slice(n, r, world)]
            assert got == list(range(n))


def _tagged(z, tag):
    pre = tag + '/'
    d = {k[len(pre):]: v for k, v in z.items() if k.startswith(pre)}
    sd = {k[2:]: torch.from_numpy(v) for k, v in d.items() if k.startswith('w/')}
    outs = {k[4:]: v for k, v in d.items() if k.startswith('out/')}
    return torch.from_numpy(d['P']), torch.from_numpy(d['F']), sd, outs


def test_ablation_models_match_reference_runs(golden_dir):
    from video_frame_inpainting_amd import ablations as ab
    z = _load(golden_dir, 'ablations.npz')
    P, Fo, sd, outs = _tagged(z, 'twi')
    m = ab.TimeWeightedInterpolationFillInModel(4, 1, 3, 7, num_block=5, kf_dim=2)
    m.load_state_dict(sd)                                   # mcnet.* / interp_net.* / merge_residual*: the reference's keys
    m.interp_net.separableConvolution = lambda i, v, h, ks: torch.from_numpy(sepconv_oracle.forward(i.numpy(), v.numpy(), h.numpy(), ks))
    with torch.no_grad():
        o = m(3, P, Fo)
    assert set(o) == set(outs)
    for k in outs:
        np.testing.assert_allclose(o[k].numpy(), outs[k], rtol=1e-4, atol=2e-5, err_msg='twi ' + k)
    for tag, cls, c_dim in (('bi_twa', ab.BidirectionalTimeWeightedAverageFillInModel, 1),
                            ('bi_sa', ab.BidirectionalSimpleAverageFillInModel, 3)):
        P, Fo, sd, outs = _tagged(z, tag)
        m = cls(4, c_dim, 3)
        m.load_state_dict(sd)
        with torch.no_grad():
            o = m(4, P, Fo)
        assert set(o) == set(outs)
        for k in outs:
            np.testing.assert_allclose(o[k].numpy(), outs[k], rtol=1e-4, atol=2e-5, err_msg=tag + ' ' + k)
    P, Fo = torch.from_numpy(z['tw_p_f/P']), torch.from_numpy(z['tw_p_f/F'])
    np.testing.assert_allclose(ab.TimeWeightedPFFillInModel()(3, P, Fo)['pred'].numpy(), z['tw_p_f/out/pred'], rtol=1e-6, atol=1e-7)
    assert isinstance(vfi.create_model('TimeWeightedInterpolationFillInModel_gray'), ab.TimeWeightedInterpolationFillInModel)
    assert isinstance(vfi.create_model('BidirectionalSimpleAverageFillInModel_color'), ab.BidirectionalSimpleAverageFillInModel)


def test_quant_eval_pipeline_on_png_frames(tmp_path):
    # predict.py's PNG layout -> compute_quant_results.py -> results.npz -> summarize_quant_results.py table
    import compute_quant_results, summarize_quant_results
    from PIL import Image
    rs = np.random.RandomState(3)
    K, T = 2, 3
    want_p = np.zeros((2, T))
    for n in range(2):
        d = tmp_path / 'qual' / ('video_%d' % n)
        os.makedirs(d)
        for t in range(K, K + T):
            gt = rs.randint(0, 256, (24, 20)).astype(np.uint8)
            pred = np.clip(gt.astype(int) + rs.randint(-6, 7, gt.shape), 0, 255).astype(np.uint8)
            Image.fromarray(gt).save(d / ('gt_middle_%04d.png' % t))
            Image.fromarray(pred).save(d / ('pred_middle_%04d.png' % t))
            want_p[n, t - K] = metrics.psnr_uint8(pred, gt)
    compute_quant_results.main([str(tmp_path / 'qual'), str(tmp_path / 'quant'), str(K), str(T)])
    z = np.load(tmp_path / 'quant' / 'results.npz')
    assert z['psnr'].shape == (2, T) and z['ssim'].shape == (2, T) and len(z['video']) == 2
    np.testing.assert_allclose(z['psnr'], want_p)
    summarize_quant_results.main([str(tmp_path / 'tables'), '--results', '%s:bi-TAI (ours)' % (tmp_path / 'quant')])
    txt = open(tmp_path / 'tables' / 'psnr_perf_summary.txt').read()
    assert 'bi-TAI (ours)' in txt and ('%.2f' % want_p.mean(axis=1).mean()) in txt and txt.startswith('+')
    with pytest.raises(RuntimeError):
        compute_quant_results.main([str(tmp_path / 'qual'), str(tmp_path / 'quant'), '5', '5'])


def test_kxk_convolution_as_blocked_3x3_over_shifted_haloed_copies():
    """The formulation behind tai_conv_shift_stack + tai_conv3x3_wino_forward_window (MotionEnc's 5x5 / 7x7 layers,
    mcnet.py:36-38, 45-47), in plain torch: S*S shifted copies with their own halo, the k x k filter cut into S x S
    blocks of 3 x 3 taps, one unpadded 3x3 convolution."""
    import torch.nn.functional as F
    from video_frame_inpainting_amd.conv_ops import _block3x3_weight
    for k in (5, 7):
        g = torch.Generator().manual_seed(k)
        x = torch.randn(2, 3, 10, 12, generator=g, dtype=torch.float64)
        w = torch.randn(4, 3, k, k, generator=g, dtype=torch.float64)
        S = (k + 2) // 3
        N, C, H, W = x.shape
        stack = torch.zeros(N, S * S * C, H + 2, W + 4, dtype=torch.float64)
        for a in range(S):
            for b in range(S):
                oy, ox = 3 * a - k // 2 + 1, 3 * b - k // 2 + 1
                for u in range(H + 2):
                    for v in range(W + 4):
                        sy, sx = u - 1 + oy, v - 2 + ox
                        if 0 <= sy < H and 0 <= sx < W:
                            stack[:, (a * S + b) * C:(a * S + b + 1) * C, u, v] = x[:, :, sy, sx]
        y = F.conv2d(stack, _block3x3_weight(w))[:, :, :, 1:W + 1]          # "valid" on the haloed plane, origin (1, 2)
        ref = F.conv2d(x, w, padding=k // 2)
        assert (y - ref).abs().max().item() < 1e-12


def test_conv_bias_act_takes_channel_parts_on_cpu():
    from video_frame_inpainting_amd.conv_ops import conv_bias_act, conv_bias_act_maxpool
    import torch.nn.functional as F
    a, b = torch.randn(2, 8, 6, 6), torch.randn(2, 8, 6, 6)
    conv = torch.nn.Conv2d(16, 4, 3, padding=1)
    with torch.no_grad():
        assert torch.equal(conv_bias_act((a, b), conv.weight, conv.bias, 1, 'relu'), torch.relu(conv(torch.cat((a, b), 1))))
        y, yp = conv_bias_act_maxpool(a, torch.nn.Conv2d(8, 4, 3, padding=1).weight, conv.bias, 1, 'relu')
        assert torch.equal(yp, F.max_pool2d(y, 2))


def test_committed_counter_summary_matches_the_library_it_would_be_quoted_for():
    """bench.py prints `roofline.traffic` from profiles/sepconv_fwd_pmc.json only when the summary was collected on the library
    version and default kernel that are about to be measured (otherwise null).  A kernel change that forgets to re-collect the
    counters should fail here, not show up as a null in the driver's record."""
    import json
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rec = json.load(open(os.path.join(root, 'profiles', 'sepconv_fwd_pmc.json')))
    src = open(os.path.join(root, 'video-frame-inpainting_amd', 'csrc', 'sepconv_capi.hip')).read()
    version = int(re.search(r'int tai_sepconv_version\(void\) \{ return (\d+); \}', src).group(1))
    default = re.search(r'return !tileable \? 1 : \(C == 1 \? (\d+) : (\d+)\);', src)
    assert rec['library_version'] == version, (rec['library_version'], version)
    assert rec['forward_variant'] == int(default.group(1))
    assert rec['shape'] == [32, 1, 128, 128] and 0.98 < rec['traffic_over_algorithmic'] < 1.05
    c3 = json.load(open(os.path.join(root, 'profiles', 'sepconv_fwd_pmc_c3.json')))
    assert c3['library_version'] == version


def test_cpu_threads_of_the_baseline_leg_follow_affinity_quota_and_request(monkeypatch, tmp_path):
    import bench
    monkeypatch.setattr(os, 'sched_getaffinity', lambda pid: set(range(256)))
    monkeypatch.delenv('TAI_CPU_THREADS', raising=False)
    real_open = open

    def fake_open(path, *a, **k):
        if path == '/sys/fs/cgroup/cpu.max':
            import io
            return io.StringIO(fake_open.content)
        return real_open(path, *a, **k)
    monkeypatch.setattr('builtins.open', fake_open)
    fake_open.content = '1600000 100000\n'
    n, how = bench.host_cpu_share()
    assert n == 16 and how['cgroup_quota_cores'] == 16 and how['sched_affinity'] == 256
    fake_open.content = 'max 100000\n'
    assert bench.host_cpu_share()[0] == 16                 # no quota: the pool's one-GPU share
    assert bench.host_cpu_share(64)[0] == 64               # an explicit request within the affinity
    assert bench.host_cpu_share(1000)[0] == 256
    fake_open.content = '400000 100000\n'
    assert bench.host_cpu_share(64)[0] == 4                # the quota wins over a request
