"""One bi-TAI training step (BASELINE configs[2] geometry: 128x128 gray, K=T=F=5, GAN + reconstruction losses) on the GPU
against the CPU oracle's training leg (oracle/train_oracle.py): every loss term of the generator and discriminator
objectives, gradients of generator parameters from the first MC-Net convolution to the last kernel-generator layer
(through the HIP sepconv backward kernels, the Winograd input-gradient path and MIOpen's weight gradients), and the
gradient of every discriminator parameter (13 windows x 2 evaluations, each renormalising the spectral-norm weights in
place).  Reduced width for the parametrised case, the full TAI_gray width (gf_dim 64, df_dim 64) for one B = 2 step."""
import numpy as np
import pytest
import torch

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.environments import TAITrainingEnvironment
from oracle import train_oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ALPHA, BETA, IP, DISC_T = 1.0, 0.02, 3, 3            # options.py:79-80 defaults; exp_args KTH Ip / disc_t

GRAD_KEYS = ('generator.motion_enc.dyn_conv1.0.weight',      # first convolution of MC-Net (5x5, one input channel)
             'generator.motion_enc.dyn_conv3.1.weight',      # 7x7
             'generator.conv_lstm_cell.conv.weight', 'generator.conv_lstm_cell.conv.bias',
             'generator.content_enc.cont_conv2.3.weight',
             'generator.dec_cnn.dec1.2.weight',               # last (transposed) convolution of MC-Net
             'merge_residual2.res.0.weight',
             'kernelnet.moduleConv.0.0.weight', 'kernelnet.moduleUpsample.3.1.weight',
             'kernelnet.moduleVertical1.7.weight', 'kernelnet.moduleHorizontal2.7.bias')
LOSS_RTOL = 2e-4          # loss terms, relative
GRAD_RTOL = 5e-3          # max |g_gpu - g_oracle| <= GRAD_RTOL * max |g_oracle| per parameter (measured: <= 3.3e-3, profiles/r02_training_parity.txt)


@pytest.fixture(autouse=True)
def _fp32_convs():
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False


def _seeded_env(tmp_path, gf_dim, kf_dim, df_dim, K, T, F, H, W):
    model = vfi.TAIFillInModel(gf_dim, 1, 3, 51, num_block=5, kf_dim=kf_dim)
    env = TAITrainingEnvironment(model, str(tmp_path), 'exp', [H, W], 1, ALPHA, BETA, 1e-4, 0.5, df_dim, IP, DISC_T, K, T, F,
                                 [0, 0], device=DEV)
    synthetic.seeded_init(env.generator, 21)           # the environment's own init has zero biases: taps ~1e-4
    synthetic.seeded_init(env.discriminator, 22)
    g = torch.Generator().manual_seed(23)
    u = {}
    for name, m in env.discriminator.named_modules():
        if hasattr(m, 'Ip'):                              # the reference draws u ~ N(0,1) on first use: fixed here
            u[name] = torch.randn(1, m.weight.size(0), generator=g)
            m.u = u[name].to(DEV)
    return env, u


def _step_and_compare(tmp_path, gf_dim, kf_dim, df_dim, B, K=5, T=5, F=5, H=128, W=128):
    env, u = _seeded_env(tmp_path, gf_dim, kf_dim, df_dim, K, T, F, H, W)
    gen_sd = {k: v.detach().cpu().clone() for k, v in env.generator.state_dict().items()}
    disc = train_oracle.DiscriminatorState({k: v.detach().cpu() for k, v in env.discriminator.state_dict().items()}, u, IP, DISC_T)
    clips = torch.from_numpy(synthetic.make_clips(B, K + T + F, 1, H, W, synthetic.SEEDS['cfg3']))
    P, GT, Fo = synthetic.split_clip(clips, K, T, F)

    # the product: the reference's step order (environments.py:348-355) without the two optimiser updates
    env.set_train_inputs(P, Fo, GT)
    env.K, env.T, env.F = K, T, F
    env.train()
    env.forward_train()
    env.optimizer_G.zero_grad()
    env.compute_loss_G()
    env.loss_G.backward()
    g_gpu = {k: p.grad.detach().cpu().clone() for k, p in env.generator.named_parameters() if k in GRAD_KEYS}
    env.optimizer_D.zero_grad()
    env.compute_loss_D()
    env.loss_D.backward()
    d_gpu = {k: p.grad.detach().cpu().clone() for k, p in env.discriminator.named_parameters()}
    errs = env.get_current_errors()

    losses, g_ref, d_ref, out_ref = train_oracle.training_step(gen_sd, disc, 1, 5, 51, P, GT, Fo, ALPHA, BETA, list(GRAD_KEYS))

    report = []
    for k in sorted(losses):
        rel = abs(errs[k] - losses[k]) / max(abs(losses[k]), 1e-12)
        report.append('%-16s gpu %.7g oracle %.7g rel %.2e' % (k, errs[k], losses[k], rel))
        assert rel <= LOSS_RTOL, report[-1]
        assert abs(losses[k]) > 1e-4, report[-1]                       # a vanishing term would make the check vacuous
    for name, got, want in [(k, g_gpu[k], g_ref[k]) for k in GRAD_KEYS] + [('D.' + k, d_gpu[k], d_ref[k]) for k in sorted(d_ref)]:
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        report.append('%-44s max|g| %.3e  err/max %.2e' % (name, scale, err / max(scale, 1e-30)))
        assert scale > 1e-7, report[-1]
        assert err <= GRAD_RTOL * scale, report[-1]
    # the discriminator's weights were renormalised 3 x 13 times in place on both sides: same end state
    for k, v in env.discriminator.state_dict().items():
        ref = disc.sd[k]
        assert float((v.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), k
    print('\n'.join(report))
    return errs


def test_training_step_cfg3_geometry_reduced_width(tmp_path):
    _step_and_compare(tmp_path, gf_dim=8, kf_dim=4, df_dim=8, B=2)


def test_training_step_full_width(tmp_path):
    """TAI_gray as configs[2] trains it (gf_dim 64, kf_dim 32, df_dim 64), B = 2."""
    _step_and_compare(tmp_path, gf_dim=64, kf_dim=32, df_dim=64, B=2)


def test_training_step_short_context_nonsquare(tmp_path):
    """sample_KTF draws K, F >= 2 and T >= 1 per step (environments.py:417-427): K != F (no direction fusion), T = 2."""
    _step_and_compare(tmp_path, gf_dim=8, kf_dim=4, df_dim=8, B=2, K=4, T=2, F=3, H=64, W=96)


@pytest.mark.gpu
@pytest.mark.parametrize('shape, Ip', [((64, 3, 4, 4), 3), ((128, 64, 4, 4), 3), ((512, 256, 4, 4), 3), ((1, 32768), 1),
                                       ((37, 5, 3, 3), 2)])
def test_spectral_norm_kernels_match_the_oracle_power_iteration(shape, Ip):
    """tai_sn_power_iteration (2 Ip + 1 launches) against the reference's matmul form restated in oracle/train_oracle.py:
    three consecutive renormalisations (the effect is cumulative and u persists), fp32 tolerance of a few ulps."""
    from video_frame_inpainting_amd.sn_discriminator import SNConv2d, SNLinear
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    layer = (SNLinear(shape[1], shape[0], Ip=Ip) if len(shape) == 2 else SNConv2d(shape[1], shape[0], shape[2], Ip=Ip)).to(dev)
    w0 = torch.randn(shape, generator=g) * 0.05
    u0 = torch.randn(1, shape[0], generator=g)
    with torch.no_grad():
        layer.weight.copy_(w0)
    layer.u = u0.to(dev)
    ref_w, ref_u = w0.double(), u0.double()
    for call in range(3):
        out = layer._renormalise_()
        sigma, ref_u = train_oracle.max_singular_value(ref_w.view(shape[0], -1), ref_u, Ip)
        ref_w = ref_w / sigma
        assert out.data_ptr() != layer.weight.data_ptr() and out.requires_grad       # a value of the moment, not the parameter
        np.testing.assert_allclose(float(layer.last_sigma), float(sigma), rtol=2e-6)
        np.testing.assert_allclose(layer.weight.detach().cpu().numpy(), ref_w.float().numpy(), rtol=5e-6, atol=1e-9)
        np.testing.assert_allclose(layer.u.cpu().numpy(), ref_u.float().numpy(), rtol=0, atol=2e-6)
        assert torch.equal(out.detach(), layer.weight.detach())
    # a second layer from the same state ends in the same bits (fixed summation order)
    layer2 = (SNLinear(shape[1], shape[0], Ip=Ip) if len(shape) == 2 else SNConv2d(shape[1], shape[0], shape[2], Ip=Ip)).to(dev)
    with torch.no_grad():
        layer2.weight.copy_(w0)
    layer2.u = u0.to(dev)
    for call in range(3):
        layer2._renormalise_()
    assert torch.equal(layer2.weight, layer.weight) and torch.equal(layer2.u, layer.u)


@pytest.mark.gpu
def test_sn_linear_single_logit_matches_linear():
    from video_frame_inpainting_amd.sn_discriminator import SNLinear
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    layer = SNLinear(4096, 1).to(dev)
    x = torch.randn(6, 4096, device=dev, requires_grad=True)
    y = layer(x)
    w = layer.weight.detach().clone()
    assert y.shape == (6, 1)
    ref = torch.nn.functional.linear(x.detach().double(), w.double(), layer.bias.detach().double())
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    y.sum().backward()
    np.testing.assert_allclose(layer.weight.grad.cpu().numpy(), x.detach().sum(0, keepdim=True).cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(x.grad.cpu().numpy(), w.expand(6, -1).cpu().numpy(), rtol=1e-6)
    assert float(layer.bias.grad) == 6.0


@pytest.mark.gpu
def test_graphed_update_matches_the_eager_update(tmp_path):
    """train_step with graph_step (two eager updates, then one captured update replayed) against the same four updates
    run eagerly on an identically initialised environment: same loss terms every update and the same weights at the
    end, to fp32 rounding (MIOpen's weight-gradient kernels accumulate with atomics: not bit-reproducible)."""
    K = T = F = 3
    H = W = 64
    B = 2
    clips = torch.from_numpy(synthetic.make_clips(4 * B, K + T + F, 1, H, W, synthetic.SEEDS['cfg3']))
    envs = {}
    history = {}
    for mode in ('eager', 'graph'):
        model = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16)
        env = TAITrainingEnvironment(model, str(tmp_path / mode), 'exp', [H, W], 1, ALPHA, BETA, 1e-3, 0.5, 16, IP, DISC_T, K, T, F,
                                     [0, 0], device=DEV, graph_step=True)
        if mode == 'eager':
            env.STEP_GRAPH_WARMUP = 10 ** 9            # same optimizer arithmetic (Adam's counter on the device), never captured
        synthetic.seeded_init(env.generator, 21)
        synthetic.seeded_init(env.discriminator, 22)
        g = torch.Generator().manual_seed(23)
        for name, m in env.discriminator.named_modules():
            if hasattr(m, 'Ip'):
                m.u = torch.randn(1, m.weight.size(0), generator=g).to(DEV)
        env.K, env.T, env.F = K, T, F
        env.train()
        history[mode] = []
        for step in range(4):
            P, GT, Fo = synthetic.split_clip(clips[step * B:(step + 1) * B], K, T, F)
            env.train_step(P, Fo, GT)
            history[mode].append(env.get_current_errors())
        envs[mode] = env
    state = envs['graph']._step_graphs
    assert len(state) == 1 and 'graph' in next(iter(state.values()))                     # updates 3 and 4 were replays
    assert all('graph' not in v for v in envs['eager']._step_graphs.values())
    for step in range(4):
        for k, v in history['eager'][step].items():
            assert abs(history['graph'][step][k] - v) <= 5e-4 * max(abs(v), 1e-3), (step, k, history['graph'][step][k], v)
    for part in ('generator', 'discriminator'):
        ref = getattr(envs['eager'], part).state_dict()
        for k, v in getattr(envs['graph'], part).state_dict().items():
            # Adam normalises every gradient element by its own running magnitude: where a gradient is at rounding level
            # (MIOpen's atomics reorder sums from run to run) the update's direction is noise, so single elements may
            # differ by up to lr x updates = 4e-3 each way (eight runs: max 1.5e-5 .. 3.4e-3, mean over a tensor <= 1.5e-5, loss terms <= 3.6e-5
            # relative); anything systematic (a stale weight, a skipped update) is far larger
            diff = (v - ref[k]).abs()
            assert float(diff.max()) <= 8e-3, (part, k, float(diff.max()))           # 2 x lr x updates: Adam's reach
            assert float(diff.mean()) <= 2e-4, (part, k, float(diff.mean()))         # a missed update moves every element by ~lr = 1e-3
    # after replays an eager forward sees the CURRENT weights (derived Winograd filters rebuilt)
    P, GT, Fo = synthetic.split_clip(clips[:B], K, T, F)
    outs = []
    for mode in ('eager', 'graph'):
        env = envs[mode]
        env.eval()
        with torch.no_grad():
            outs.append(env.generator(T, P.to(DEV), Fo.to(DEV))['pred'])
    assert float((outs[0] - outs[1]).abs().max()) <= 5e-3
    # ... and not the weights of the update before (what the captured Winograd filters were computed from)
    assert torch.isfinite(outs[1]).all()


# ---- reference-run pins (tests/golden/sn_disc.npz: the reference's own SNDiscriminator.py run on CPU by make_golden.py) ----
def _sn_group(z, prefix):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


@pytest.mark.gpu
@pytest.mark.parametrize('Ip', [1, 3])
def test_sn_power_iteration_kernel_matches_reference_run(golden_dir, Ip):
    """tai_sn_power_iteration against max_singular_value of the reference (SNDiscriminator.py:10-25) with a given u."""
    import os
    from video_frame_inpainting_amd.sn_discriminator import SNLinear
    z = np.load(os.path.join(golden_dir, 'sn_disc.npz'))
    W, u = torch.from_numpy(z['msv_ip%d/W' % Ip]), torch.from_numpy(z['msv_ip%d/u' % Ip])
    layer = SNLinear(W.shape[1], W.shape[0], Ip=Ip).to(DEV)
    with torch.no_grad():
        layer.weight.copy_(W)
    layer.u = u.to(DEV)
    layer._renormalise_()
    sigma = z['msv_ip%d/sigma' % Ip].reshape(-1)[0]
    np.testing.assert_allclose(float(layer.last_sigma), float(sigma), rtol=2e-6)
    np.testing.assert_allclose(layer.u.cpu().numpy(), z['msv_ip%d/u_out' % Ip], rtol=0, atol=2e-6)
    np.testing.assert_allclose(layer.weight.detach().cpu().numpy(), (W / float(sigma)).numpy(), rtol=5e-6, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize('tag, c_dim', [('disc_gray', 1), ('disc_color', 3)])
def test_gpu_discriminator_matches_reference_run_over_two_calls(golden_dir, tag, c_dim):
    """The product's all-windows-in-one-pass discriminator (renormalisations up front, one factor per window) against
    the reference's window-by-window SNDiscriminator.forward (:140-159) over two consecutive calls: logits, the
    in-place renormalised weights and the persistent u vectors."""
    import os
    from video_frame_inpainting_amd.sn_discriminator import SNDiscriminator
    z = np.load(os.path.join(golden_dir, 'sn_disc.npz'))
    w0, u0 = _sn_group(z, tag + '/w0/'), _sn_group(z, tag + '/u0/')
    disc = SNDiscriminator((32, 32), c_dim, 3, 4, 3)
    disc.load_state_dict(w0)
    disc.to(DEV)
    mods = dict(disc.named_modules())
    for name, u in u0.items():
        mods[name].u = u.to(DEV)
    for call in range(2):
        frames = torch.from_numpy(z['%s/frames%d' % (tag, call)]).to(DEV)
        want = z['%s/logits%d' % (tag, call)]
        with torch.no_grad():
            got = disc(frames)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-4, atol=2e-5 * float(np.abs(want).max()))
        for k, v in _sn_group(z, '%s/w%d/' % (tag, call + 1)).items():
            np.testing.assert_allclose(disc.state_dict()[k].cpu().numpy(), v.numpy(), rtol=2e-5, atol=1e-7, err_msg=k)
        for k, v in _sn_group(z, '%s/u%d/' % (tag, call + 1)).items():
            np.testing.assert_allclose(mods[k].u.cpu().numpy(), v.numpy(), rtol=0, atol=5e-6, err_msg=k)


@pytest.mark.gpu
def test_gpu_discriminator_gradients_match_reference_run_single_window(golden_dir):
    import os
    import torch.nn.functional as Fn
    from video_frame_inpainting_amd.sn_discriminator import SNDiscriminator
    z = np.load(os.path.join(golden_dir, 'sn_disc.npz'))
    w0, u0 = _sn_group(z, 'disc_grad/w0/'), _sn_group(z, 'disc_grad/u0/')
    want = _sn_group(z, 'disc_grad/grad/')
    disc = SNDiscriminator((32, 32), 1, 3, 4, 3)
    disc.load_state_dict(w0)
    disc.to(DEV)
    mods = dict(disc.named_modules())
    for name, u in u0.items():
        mods[name].u = u.to(DEV)
    frames = torch.from_numpy(z['disc_grad/frames']).to(DEV).requires_grad_(True)
    logits = disc(frames)
    loss = Fn.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), z['disc_grad/logits'], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(float(loss.detach()), float(z['disc_grad/loss'][0]), rtol=1e-5)
    for k, p in disc.named_parameters():
        scale = float(want[k].abs().max())
        assert scale > 1e-8, k
        assert float((p.grad.cpu() - want[k]).abs().max()) <= 1e-3 * scale, (k, float((p.grad.cpu() - want[k]).abs().max()), scale)
    gf = z['disc_grad/grad_frames']
    assert float(np.abs(frames.grad.cpu().numpy() - gf).max()) <= 1e-3 * float(np.abs(gf).max())
