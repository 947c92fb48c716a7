"""bi-TAI on the GPU: the product model (HIP sepconv + MIOpen convs) against the golden whole-model runs of the
reference, against the CPU oracle on a synthetic clip (output and PSNR/SSIM parity), hipGraph replay, and one
training step of the TAI training environment."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import metrics, synthetic
from video_frame_inpainting_amd.environments import TAITrainingEnvironment, create_eval_environment
from video_frame_inpainting_amd.graph import GraphedForward
from oracle import tai_oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
KEYS = ('pred', 'pred_forward', 'pred_backward', 'interp_net_outputs_1', 'interp_net_outputs_2')


@pytest.fixture(autouse=True)
def _fp32_convs():
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('tag', ['gray', 'color'])
def test_matches_reference_golden_run(golden_dir, tag):
    z = _load(golden_dir, 'tai_%s.npz' % tag)
    sd = {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith('w/')}
    m = vfi.TAIFillInModel(4, int(z['c_dim'][0]), 3, int(z['ks'][0]), num_block=int(z['num_block'][0]), kf_dim=2)
    m.load_state_dict(sd)
    m.to(DEV).eval()
    with torch.no_grad():
        out = m(int(z['T'][0]), torch.from_numpy(z['P']).to(DEV), torch.from_numpy(z['F']).to(DEV))
    for k in KEYS:                      # full model, tanh-bounded outputs: max-abs 2e-4
        np.testing.assert_allclose(out[k].cpu().numpy(), z['out/' + k], rtol=0, atol=2e-4, err_msg=k)


REL_TOL = 1e-4      # per output key: max |gpu - oracle| <= REL_TOL * max |oracle|  (fp32, different summation orders)


def _assert_matches_oracle(out, ref, GT, name='', min_levels=50, min_spread=0.05, rel_tol=REL_TOL):
    """Outputs within REL_TOL of the oracle relative to each key's own magnitude, AND the comparison is about something:
    the uint8 prediction spans many gray levels, PSNR vs ground truth differs between frames, and PSNR / SSIM vs ground
    truth agree between GPU and oracle (0.01 dB / 1e-4)."""
    errs = {}
    for k in KEYS:
        scale = float(ref[k].abs().max())
        assert scale > 0.05, (name, k, scale)                 # a near-zero reference would make the check vacuous
        err = float((out[k].cpu() - ref[k]).abs().max())
        errs[k] = err / scale
        assert err <= rel_tol * scale, (name, k, err, scale)
    print('%s: max |gpu - oracle| / max |oracle|  %s  (bound %.0e)' % (name, '  '.join('%s %.2e' % kv for kv in errs.items()), rel_tol))      # pytest -s
    pred_gpu, pred_cpu = out['pred'].cpu().numpy(), ref['pred'].numpy()
    assert len(np.unique(metrics.to_uint8(pred_gpu))) > min_levels, name
    p_gpu, s_gpu, _ = metrics.compute_errors(pred_gpu, GT.numpy())
    p_cpu, s_cpu, _ = metrics.compute_errors(pred_cpu, GT.numpy())
    if p_cpu.size > 1:
        assert p_cpu.max() - p_cpu.min() > min_spread, (name, p_cpu)       # frames differ: the deltas below carry information
    assert np.max(np.abs(p_gpu - p_cpu)) <= 0.01, (name, p_gpu, p_cpu)     # dB
    assert np.max(np.abs(s_gpu - s_cpu)) <= 1e-4, (name, s_gpu, s_cpu)


def test_cfg1_shape_matches_cpu_oracle_and_psnr_parity():
    # BASELINE config 0 geometry (TAI_gray, 128x128 gray, K=F=5, T=5, B=1) at reduced width so the CPU side takes seconds;
    # seeded weights AND biases (zero biases give taps ~1e-4 and constant-gray predictions: no evidence)
    m = synthetic.seeded_init(vfi.TAIFillInModel(8, 1, 3, 51, num_block=5, kf_dim=4), 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    clips = synthetic.make_clips(1, 15, 1, 128, 128, synthetic.SEEDS['cfg1'])
    P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, 5, 5))
    with torch.no_grad():
        ref = tai_oracle.tai_forward(sd, 1, 5, 51, 5, P, Fo)
        out = m.to(DEV).eval()(5, P.to(DEV), Fo.to(DEV))
    _assert_matches_oracle(out, ref, GT, 'cfg1 shape')


def _at_the_timed_batch_dispatch(monkeypatch, scale, m, T, P, Fo, ref, GT, name, rel_tol=REL_TOL):
    """The same clips once more with every convolution routed as in the timed batch of the config (tests/conftest.py: DispatchAt;
    ``scale`` x these clips), with F(4x4, 3x3) on the wide layers (the default) and with F(2x2, 3x3) everywhere: the wide
    low-resolution layers that a 1-2 clip batch hands to MIOpen then run the in-tree kernels the bench times.  No ATen convolution
    may be left in the forward."""
    from conftest import DispatchAt, miopen_convolutions
    from video_frame_inpainting_amd import conv_ops
    with monkeypatch.context() as mp:
        d = DispatchAt(mp, scale)
        for tile in (4, 2):
            prev = conv_ops.set_winograd_tile(tile)
            try:
                with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True) as prof:
                    out = m(T, P.to(DEV), Fo.to(DEV))
            finally:
                conv_ops.set_winograd_tile(prev)
            _assert_matches_oracle(out, ref, GT, '%s, dispatch of %d x the clips, Winograd tile %d' % (name, scale, tile), rel_tol=rel_tol)
            assert not miopen_convolutions(prof), miopen_convolutions(prof)[:4]
        assert d.routes['wino43'] > 0, d.routes


def test_full_width_tai_gray_matches_cpu_oracle(monkeypatch):
    """The model BASELINE's headline is quoted on -- TAI_gray, gf_dim 64, 128x128, K=F=T=5 -- against the CPU oracle on two
    clips: every 3x3 layer runs on the Winograd-MFMA kernels at their production shapes (the reduced-width tests fall
    below conv_ops.WINO_MIN_WORKGROUPS on most layers), the 5x5 / 7x7 layers on the shifted-stack path."""
    m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    clips = synthetic.make_clips(2, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
    P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, 5, 5))
    with torch.no_grad():
        ref = tai_oracle.tai_forward(sd, 1, 5, 51, 5, P, Fo)
        m.to(DEV).eval()
        out = m(5, P.to(DEV), Fo.to(DEV))
        _assert_matches_oracle(out, ref, GT, 'full width, eager')
        g = GraphedForward(m, 5, P.to(DEV), Fo.to(DEV))
        _assert_matches_oracle(g(), ref, GT, 'full width, hipGraph replay')
        _at_the_timed_batch_dispatch(monkeypatch, 16, m, 5, P, Fo, ref, GT, 'configs[1]')


def test_full_width_tai_color_matches_cpu_oracle(monkeypatch):
    """configs[3]'s model -- create_model('TAI_color') = TAIFillInModel(64, 3, 3, 51, num_block=4) (create_model.py:29-30),
    256x256 BGR, K = F = 3, T = 5 -- at its real width against the CPU oracle on one clip: the 256^2 Winograd layer
    shapes, the three-channel sepconv kernel (sepconv_forward_asm_c3) at [5,3,256,256], the decoder that never injects
    the time ratio (tai.py:213-217 with num_block 4), the BGR -> gray temporal differences."""
    m = synthetic.seeded_init(vfi.create_model('TAI_color'), 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    clips = synthetic.make_clips(1, 11, 3, 256, 256, synthetic.SEEDS['cfg4'])
    P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 3, 5, 3))
    with torch.no_grad():
        ref = tai_oracle.tai_forward(sd, 3, 4, 51, 5, P, Fo)
        m.to(DEV).eval()
        out = m(5, P.to(DEV), Fo.to(DEV))
        _assert_matches_oracle(out, ref, GT, 'TAI_color full width, eager')
        g = GraphedForward(m, 5, P.to(DEV), Fo.to(DEV))
        _assert_matches_oracle(g(), ref, GT, 'TAI_color full width, hipGraph replay')
        _at_the_timed_batch_dispatch(monkeypatch, 16, m, 5, P, Fo, ref, GT, 'configs[3]')


def test_full_width_tai_gray_long_gap_matches_cpu_oracle(monkeypatch):
    """configs[4]: TAI_gray at full width with T = 10 middle frames (alt_T), K = F = 5, one clip."""
    m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    clips = synthetic.make_clips(1, 20, 1, 128, 128, synthetic.SEEDS['cfg5'])
    P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, 5, 10, 5))
    with torch.no_grad():
        ref = tai_oracle.tai_forward(sd, 1, 5, 51, 10, P, Fo)
        m.to(DEV).eval()
        out = m(10, P.to(DEV), Fo.to(DEV))
        assert out['pred'].shape == (1, 10, 1, 128, 128)
        # ten recurrent MC-Net steps instead of five: the tanh-bounded predictions drift to 0.8-1.0e-4 of their maximum on some
        # clips (bench.py's configs[4] check) -- inside SURVEY.md 8d's 2e-4 bound for the full model, which is the bound used here
        _assert_matches_oracle(out, ref, GT, 'TAI_gray T=10 full width', rel_tol=2e-4)
        _at_the_timed_batch_dispatch(monkeypatch, 32, m, 10, P, Fo, ref, GT, 'configs[4]', rel_tol=2e-4)


def test_headline_forward_is_bit_reproducible():
    """configs[1]'s forward (TAI_gray, 32 clips, hipGraph replay) takes in-tree kernels with fixed summation orders for every layer
    -- no MIOpen kernel with atomics is left in it -- and the persistent sepconv kernel's LDS-counter synchronisation must not let
    a stale read through: replays and an eager pass all give the same bits (tools/soak_forward.py runs thousands)."""
    m = synthetic.seeded_init(vfi.create_model('TAI_gray'), 0).to(DEV).eval()
    clips = synthetic.make_clips(32, 15, 1, 128, 128, synthetic.SEEDS['cfg2'])
    P, _, Fo = (torch.from_numpy(x).to(DEV) for x in synthetic.split_clip(clips, 5, 5, 5))
    with torch.no_grad():
        eager = {k: v.clone() for k, v in m(5, P, Fo).items()}
        g = GraphedForward(m, 5, P, Fo, warmup=1)
        for _ in range(5):
            out = g()
            for k in KEYS:
                assert torch.equal(out[k], eager[k]), k


def test_derived_weights_follow_in_place_weight_writes():
    """conv_ops caches Winograd-domain / flipped weights per tensor version: a write that moves the version
    counter is seen by itself; a write through .data needs conv_ops.invalidate_derived (weights_init, the replica
    broadcast and the environments' load call it)."""
    from video_frame_inpainting_amd import conv_ops
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV)
    convt = torch.nn.ConvTranspose2d(64, 64, 3, padding=1).to(DEV)
    x = torch.randn(8, 64, 64, 64, device=DEV)

    def check():
        with torch.no_grad():
            got = conv_ops.conv_bias_act(x, conv.weight, conv.bias, 1, 'relu')
            want = torch.relu(F.conv2d(x, conv.weight, conv.bias, padding=1))
            assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
            got = conv_ops.conv_bias_act(x, convt.weight, convt.bias, 1, None, transposed=True)
            want = F.conv_transpose2d(x, convt.weight, convt.bias, padding=1)
            assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    check()
    with torch.no_grad():
        conv.weight.mul_(2.0)
        convt.weight.add_(0.01)
    check()                                     # version counter moved: rebuilt without help
    conv.weight.data.mul_(-0.5)                 # neither the version counter nor the pointer moves
    convt.weight.data.mul_(3.0)
    conv_ops.invalidate_derived(conv)
    conv_ops.invalidate_derived(convt)
    check()
    conv.apply(vfi.util.weights_init)           # the package's own re-initialisation is seen without an explicit call
    check()


def test_direction_fusion_and_graph_replay_change_nothing():
    torch.manual_seed(1)
    m = vfi.TAIFillInModel(8, 3, 3, 51, num_block=4, kf_dim=4).to(DEV).eval()
    clips = torch.from_numpy(synthetic.make_clips(2, 8, 3, 32, 32, 5)).to(DEV)
    P, Fo = clips[:, :3], clips[:, 5:]
    with torch.no_grad():
        fused = {k: v.clone() for k, v in m(2, P, Fo).items()}
        m.fuse_directions = m.batch_time_steps = False
        split = m(2, P, Fo)
        m.fuse_directions = m.batch_time_steps = True
    for k in KEYS:
        assert float((fused[k] - split[k]).abs().max()) <= 2e-5, k
    g = GraphedForward(m, 2, P, Fo)
    out = g(P, Fo)
    for k in KEYS:
        assert float((out[k] - fused[k]).abs().max()) <= 2e-5, k
    out2 = g(P.flip(0), Fo.flip(0))                 # new inputs through the same graph
    assert float((out2['pred'] - fused['pred'].flip(0)).abs().max()) <= 2e-5


def test_eval_environment_protocol(tmp_path):
    torch.manual_seed(2)
    model = vfi.TAIFillInModel(4, 1, 3, 51, num_block=5, kf_dim=2)
    os.makedirs(tmp_path / 'exp')
    torch.save({'generator': model.state_dict()}, tmp_path / 'exp' / 'model_best.ckpt')
    fresh = vfi.TAIFillInModel(4, 1, 3, 51, num_block=5, kf_dim=2)
    env = create_eval_environment(fresh, str(tmp_path), 'exp', 'model_best.ckpt', [0, 0], device=DEV, use_graph=True)
    clips = torch.from_numpy(synthetic.make_clips(2, 7, 1, 32, 32, 3))
    env.set_test_inputs(clips[:, :2], clips[:, 5:])
    env.T = 3
    env.eval()
    env.forward_test()
    with torch.no_grad():
        want = model.to(DEV).eval()(3, clips[:, :2].to(DEV), clips[:, 5:].to(DEV))['pred']
    assert env.gen_output['pred'].shape == (2, 3, 1, 32, 32)
    assert float((env.gen_output['pred'] - want).abs().max()) <= 2e-5
    with pytest.raises(RuntimeError):
        create_eval_environment(fresh, str(tmp_path), 'exp', 'missing.ckpt', [0, 0], device=DEV)


def test_one_training_step_of_the_tai_environment(tmp_path):
    torch.manual_seed(3)
    np.random.seed(0)
    model = vfi.TAIFillInModel(4, 1, 3, 51, num_block=5, kf_dim=2)
    env = TAITrainingEnvironment(model, str(tmp_path), 'exp', [32, 32], 1, 1.0, 0.02, 1e-4, 0.5, 8, 3, 3, 3, 3, 3, [0, 0],
                                 device=DEV)
    env.sync_replicas()
    clips = torch.from_numpy(synthetic.make_clips(2, 9, 1, 32, 32, 4))
    K, T, F = env.sample_KTF(False)
    env.set_train_inputs(clips[:, :K], clips[:, K + T:], clips[:, K:K + T])
    env.K, env.T, env.F = K, T, F
    before = {k: v.clone() for k, v in env.generator.state_dict().items()}
    env.train()
    env.forward_train()
    env.optimize_parameters()
    errs = env.get_current_errors()
    assert set(errs) == {'G_loss', 'G_Lp', 'G_gdl', 'D_real', 'D_fake', 'G_GAN', 'G_Lp_forward', 'G_gdl_forward',
                         'G_Lp_backward', 'G_gdl_backward'}
    assert all(np.isfinite(v) for v in errs.values())
    want = errs['G_Lp'] + errs['G_gdl'] + 0.02 * errs['G_GAN'] + errs['G_Lp_forward'] + errs['G_Lp_backward'] + \
        errs['G_gdl_forward'] + errs['G_gdl_backward']
    assert abs(errs['G_loss'] - want) <= 1e-5 * (1 + abs(want))           # environments.py:379,453
    after = env.generator.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert any(k.startswith('kernelnet.moduleVertical1') for k in moved) and any(k.startswith('generator.') for k in moved)
    assert not any(k.startswith('merge_residual1') for k in moved)        # never evaluated, as in the reference
    env.save('model_latest.ckpt', 1, 0, 0)
    snap = torch.load(tmp_path / 'exp' / 'model_latest.ckpt', weights_only=False)
    assert set(snap) == {'updates', 'sum_avg_psnr_err', 'sum_avg_ssim_err', 'generator', 'optimizer_G', 'discriminator',
                         'optimizer_D'}


def test_ablation_models_on_gpu_match_reference_runs(golden_dir):
    from video_frame_inpainting_amd import ablations as ab
    z = _load(golden_dir, 'ablations.npz')

    def tagged(tag):
        pre = tag + '/'
        d = {k[len(pre):]: v for k, v in z.items() if k.startswith(pre)}
        return (torch.from_numpy(d['P']).to(DEV), torch.from_numpy(d['F']).to(DEV),
                {k[2:]: torch.from_numpy(v) for k, v in d.items() if k.startswith('w/')},
                {k[4:]: v for k, v in d.items() if k.startswith('out/')})
    for tag, model, T in (('twi', ab.TimeWeightedInterpolationFillInModel(4, 1, 3, 7, num_block=5, kf_dim=2), 3),
                          ('bi_twa', ab.BidirectionalTimeWeightedAverageFillInModel(4, 1, 3), 4),
                          ('bi_sa', ab.BidirectionalSimpleAverageFillInModel(4, 3, 3), 4)):
        P, Fo, sd, outs = tagged(tag)
        model.load_state_dict(sd)
        model.to(DEV).eval()
        with torch.no_grad():
            o = model(T, P, Fo)
        for k in outs:
            np.testing.assert_allclose(o[k].cpu().numpy(), outs[k], rtol=0, atol=2e-4, err_msg=tag + ' ' + k)


def _tiny(c_dim, num_block, seed):
    m = synthetic.seeded_init(vfi.TAIFillInModel(8, c_dim, 3, 51, num_block=num_block, kf_dim=4), seed)
    return m, {k: v.clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize('name,c_dim,num_block,H,W,K,T,F', [
    ('cfg4-shape: colour 256x256, K=F=3, T=5', 3, 4, 256, 256, 3, 5, 3),
    ('cfg5-shape: long gap T=10', 1, 5, 128, 128, 5, 10, 5),
    ('minimum context K=F=2, T=1, non-square', 1, 5, 64, 96, 2, 1, 2),
    ('K != F: the two directions cannot be fused', 3, 4, 32, 48, 4, 3, 2),
])
def test_other_baseline_config_shapes_match_cpu_oracle(name, c_dim, num_block, H, W, K, T, F):
    m, sd = _tiny(c_dim, num_block, 11)
    clips = synthetic.make_clips(1, K + T + F, c_dim, H, W, 77)
    P, GT, Fo = (torch.from_numpy(x) for x in synthetic.split_clip(clips, K, T, F))
    with torch.no_grad():
        ref = tai_oracle.tai_forward(sd, c_dim, num_block, 51, T, P, Fo)
        out = m.to(DEV).eval()(T, P.to(DEV), Fo.to(DEV))
    for k in KEYS:
        assert out[k].shape == ref[k].shape == (1, T, c_dim, H, W)
    # width 8 gives a narrower output range than the real model, and the 4-block colour model never sees the time ratio
    # (tai.py:213-217), so its frames differ less
    _assert_matches_oracle(out, ref, GT, name, min_levels=30, min_spread=0.01)


def test_predict_and_train_drivers_run_end_to_end(tmp_path, monkeypatch):
    import predict
    import train
    monkeypatch.chdir(tmp_path)
    spec = '{"class": "TAIFillInModel", "args": [4, 1, 3, 51], "kwargs": {"num_block": 5, "kf_dim": 2}}'
    common = ['--name', 'drv', '--K', '3', '--T', '2', '--F', '3', '--c_dim', '1', '--image_size', '32', '--model_key', spec,
              '--checkpoints_dir', str(tmp_path / 'ckpt')]
    train.main(common + ['--batch_size', '2', '--max_iter', '2', '--synthetic', '4', '--print_freq', '1', '--df_dim', '8'])
    assert (tmp_path / 'ckpt' / 'drv' / 'model_latest.ckpt').exists()
    # predict.py loads what train.py saved (snapshot['generator']) and writes the reference's PNG set
    predict.main(common + ['--batch_size', '2', '--synthetic', '3', '--qual_result_root', str(tmp_path / 'res'),
                           '--snapshot_file_name', 'model_latest.ckpt', '--intermediate_preds'])
    files = sorted(os.listdir(tmp_path / 'res' / 'synthetic_000000'))
    want = (['gt_preceding_%04d.png' % i for i in range(3)] + ['gt_middle_%04d.png' % i for i in (3, 4)] +
            ['gt_following_%04d.png' % i for i in (5, 6, 7)] +
            ['%s_%04d.png' % (p, i) for p in ('pred_middle', 'pred_middle_forward', 'pred_middle_backward',
                                                'interp_net_outputs_1', 'interp_net_outputs_2') for i in (3, 4)])
    assert files == sorted(want)
    from PIL import Image
    im = np.asarray(Image.open(tmp_path / 'res' / 'synthetic_000002' / 'pred_middle_0003.png'))
    assert im.shape == (32, 32) and im.dtype == np.uint8


def test_predict_on_reference_style_video_lists(tmp_path, monkeypatch):
    """predict.py on the reference's list files (contiguous and --disjoint_clips), videos = directories of PNG frames."""
    import predict
    from PIL import Image
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(3)
    for name in ('vidA', 'vidB'):
        os.makedirs(tmp_path / name)
        base = rng.randint(0, 256, (40, 48, 3)).astype(np.uint8)
        for t in range(10):
            Image.fromarray(np.roll(base, 2 * t, axis=1)).save(tmp_path / name / ('%03d.png' % t))
    (tmp_path / 'contig.txt').write_text('%s 1-8\n%s 2-9\n' % (tmp_path / 'vidA', tmp_path / 'vidB'))
    (tmp_path / 'disjoint.txt').write_text('%s 1-3 6-8\n' % (tmp_path / 'vidA'))
    spec = '{"class": "TAIFillInModel", "args": [4, 1, 3, 51], "kwargs": {"num_block": 5, "kf_dim": 2}}'
    common = ['--name', 'lst', '--K', '3', '--T', '2', '--F', '3', '--c_dim', '1', '--image_size', '32', '--model_key', spec,
              '--checkpoints_dir', str(tmp_path / 'ckpt'), '--batch_size', '2', '--random_init']
    predict.main(common + ['--test_video_list_path', str(tmp_path / 'contig.txt'), '--qual_result_root', str(tmp_path / 'r1')])
    assert sorted(os.listdir(tmp_path / 'r1')) == ['vidA_1-8', 'vidB_2-9']
    files = sorted(os.listdir(tmp_path / 'r1' / 'vidB_2-9'))
    assert files == sorted(['gt_preceding_%04d.png' % i for i in range(3)] + ['gt_middle_%04d.png' % i for i in (3, 4)] +
                           ['gt_following_%04d.png' % i for i in (5, 6, 7)] + ['pred_middle_%04d.png' % i for i in (3, 4)])
    # the ground-truth PNG is the resized gray frame: frame 2 (1-indexed) of vidB resized 40x48 -> 32x32
    from video_frame_inpainting_amd import data as vdata
    src = np.asarray(Image.open(tmp_path / 'vidB' / '001.png'))
    bgr = vdata.resize_bilinear(src, 32, 32)[:, :, ::-1].astype(np.float32) / 255 * 2 - 1
    gray = 0.1140 * bgr[:, :, 0] + 0.5870 * bgr[:, :, 1] + 0.2989 * bgr[:, :, 2]
    want = (255 * (np.clip(gray, -1, 1) + 1) / 2).astype(np.uint8)
    got = np.asarray(Image.open(tmp_path / 'r1' / 'vidB_2-9' / 'gt_preceding_0000.png'))
    assert got.shape == (32, 32) and np.abs(got.astype(int) - want.astype(int)).max() <= 1
    predict.main(common + ['--test_video_list_path', str(tmp_path / 'disjoint.txt'), '--disjoint_clips',
                           '--qual_result_root', str(tmp_path / 'r2')])
    files = sorted(os.listdir(tmp_path / 'r2' / 'vidA_1-3_6-8'))
    assert files == sorted(['gt_preceding_%04d.png' % i for i in range(3)] + ['gt_following_%04d.png' % i for i in (5, 6, 7)] +
                           ['pred_middle_%04d.png' % i for i in (3, 4)])       # no gt_middle for disjoint clips


def test_train_driver_on_a_video_list(tmp_path, monkeypatch):
    import train
    from PIL import Image
    monkeypatch.chdir(tmp_path)
    rng = np.random.RandomState(5)
    lines = []
    for name in ('v0', 'v1', 'v2'):
        os.makedirs(tmp_path / name)
        base = rng.randint(0, 256, (32, 32, 3)).astype(np.uint8)
        for t in range(12):
            Image.fromarray(np.roll(base, t, axis=0)).save(tmp_path / name / ('%03d.png' % t))
        lines.append('%s 1-12' % (tmp_path / name))
    (tmp_path / 'train.txt').write_text('\n'.join(lines) + '\n')
    spec = '{"class": "TAIFillInModel", "args": [4, 1, 3, 51], "kwargs": {"num_block": 5, "kf_dim": 2}}'
    train.main(['--name', 'lst', '--K', '3', '--T', '2', '--F', '3', '--c_dim', '1', '--image_size', '32', '--model_key', spec,
                '--checkpoints_dir', str(tmp_path / 'ckpt'), '--batch_size', '2', '--max_iter', '2', '--print_freq', '1',
                '--df_dim', '8', '--train_video_list_path', str(tmp_path / 'train.txt'), '--num_threads', '0'])
    assert (tmp_path / 'ckpt' / 'lst' / 'model_latest.ckpt').exists()


@pytest.mark.gpu
@pytest.mark.parametrize('paths', ['both', 'h_only', 'c_only'])
def test_convlstm_gates_training_form_matches_aten(paths):
    """_LstmGates (tai_convlstm_gates_forward / _backward) against the reference's expression (mcnet.py:287-293) under fp64
    autograd: values, and the gradients of the gates tensor and of c from either or both outputs."""
    from video_frame_inpainting_amd.mcnet import _LstmGates
    g = torch.Generator().manual_seed(12)
    N, F_, H, W = 3, 16, 6, 10
    gates = torch.randn(N, 4 * F_, H, W, generator=g).cuda().requires_grad_(True)
    c = torch.randn(N, F_, H, W, generator=g).cuda().requires_grad_(True)
    gc, gh = torch.randn(N, F_, H, W, generator=g).cuda(), torch.randn(N, F_, H, W, generator=g).cuda()
    new_c, new_h = _LstmGates.apply(gates, c, 1.0)
    gd, cd = gates.detach().double().requires_grad_(True), c.detach().double().requires_grad_(True)
    i, j, f, o = torch.chunk(gd, 4, dim=1)
    rc = cd * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
    rh = torch.tanh(rc) * torch.sigmoid(o)
    assert float((new_c.double() - rc).abs().max()) <= 2e-6 and float((new_h.double() - rh).abs().max()) <= 2e-6
    loss = lambda a, b: ((a * gc.to(a.dtype)).sum() if paths != 'h_only' else 0) + ((b * gh.to(b.dtype)).sum() if paths != 'c_only' else 0)
    dg, dc = torch.autograd.grad(loss(new_c, new_h), (gates, c))
    rg, rcg = torch.autograd.grad(loss(rc, rh), (gd, cd))
    assert float((dg.double() - rg).abs().max()) <= 5e-6 and float((dc.double() - rcg).abs().max()) <= 5e-6
