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
GRAD_RTOL = 5e-3          # generator: max |g_gpu - g_oracle| <= GRAD_RTOL * max |g_oracle| per parameter
D_GRAD_RTOL = 2e-5        # discriminator, the oracle differentiating on the product's side of every LeakyReLU kink (_KinkSides): measured <= 6.6e-7
                          # (profiles/r05_training_parity.txt; with the oracle's own sides 3.3e-3 -- one element, tests/test_train_oracle_cpu.py)


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


_ORACLE = {}


def _oracle_generator_leg(key, gen_sd, disc_sd, u, P, GT, Fo):
    """The CPU oracle's G half for one test geometry, computed once per process (seeded inputs: the same key means the same step):
    (loss terms, generator gradients, fake clip, the discriminator's weights and u vectors after the evaluation inside the G loss)."""
    if key not in _ORACLE:
        disc = train_oracle.DiscriminatorState(disc_sd, u, IP, DISC_T)
        terms, g_ref, _, fake = train_oracle.generator_leg(gen_sd, disc, 1, 5, 51, P, GT, Fo, ALPHA, BETA, list(GRAD_KEYS))
        _ORACLE[key] = (terms, g_ref, fake, {k: v.clone() for k, v in disc.sd.items()}, {k: v.clone() for k, v in disc.u.items()})
    return _ORACLE[key]


class _KinkSides(object):
    """Records, for every discriminator evaluation of the product, which side of LeakyReLU's kink each pre-activation fell on
    (the sign of the layer's output: LeakyReLU keeps it).  The oracle's D half then differentiates the same piecewise-linear
    function (train_oracle.DiscriminatorState.forward, ``masks``): one element of conv_layers.0 at rounding distance from zero
    otherwise decides 3e-3 of that layer's weight gradient (tests/test_train_oracle_cpu.py)."""

    def __init__(self, monkeypatch):
        from video_frame_inpainting_amd import sn_discriminator as snd
        self.layers = []
        orig = snd._WindowScaledConvLReLU.apply

        def spy(*args):
            y = orig(*args)
            self.layers.append((y.detach() > 0).cpu())
            return y
        monkeypatch.setattr(snd._WindowScaledConvLReLU, 'apply', staticmethod(spy))

    def masks_of_call(self, call, B):
        """{(window, layer key): bool [B, Co, H, W]} of the ``call``-th discriminator evaluation (4 layers each, windows along the batch)."""
        out = {}
        for li, key in enumerate(train_oracle.SN_CONV_KEYS):
            m = self.layers[4 * call + li]
            for t0 in range(m.shape[0] // B):
                out[(t0, key)] = m[t0 * B:(t0 + 1) * B]
        return out


def _step_and_compare(tmp_path, monkeypatch, gf_dim, kf_dim, df_dim, B, K=5, T=5, F=5, H=128, W=128, profile=False, tag=''):
    env, u = _seeded_env(tmp_path, gf_dim, kf_dim, df_dim, K, T, F, H, W)
    gen_sd = {k: v.detach().cpu().clone() for k, v in env.generator.state_dict().items()}
    disc_sd = {k: v.detach().cpu().clone() for k, v in env.discriminator.state_dict().items()}
    clips = torch.from_numpy(synthetic.make_clips(B, K + T + F, 1, H, W, synthetic.SEEDS['cfg3']))
    P, GT, Fo = synthetic.split_clip(clips, K, T, F)
    sides = _KinkSides(monkeypatch)

    # the product: the reference's step order (environments.py:348-355) without the two optimiser updates
    import contextlib
    prof = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True) if profile \
        else contextlib.nullcontext()
    with prof:
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
    assert len(sides.layers) == 12, len(sides.layers)           # D(fake) in the G loss, D(fake.detach()), D(real): 4 layers each

    terms, g_ref, fake, sd1, u1 = _oracle_generator_leg((gf_dim, kf_dim, df_dim, B, K, T, F, H, W), gen_sd, disc_sd, u, P, GT, Fo)
    disc = train_oracle.DiscriminatorState(sd1, u1, IP, DISC_T)
    d_terms, d_ref = train_oracle.discriminator_leg(disc, fake, P, GT, Fo, sides.masks_of_call(1, B), sides.masks_of_call(2, B))
    # ... and with the oracle's own rounding deciding the sides, for the record (what the bound had to absorb before)
    disc_own = train_oracle.DiscriminatorState(sd1, u1, IP, DISC_T)
    _, d_own = train_oracle.discriminator_leg(disc_own, fake, P, GT, Fo)
    losses = {k: float(v) for k, v in list(terms.items()) + list(d_terms.items())}

    report = ['== %s gf_dim %d B %d K %d T %d F %d %dx%d' % (tag, gf_dim, B, K, T, F, H, W)]
    worst = {'G': 0.0, 'D': 0.0, 'D own sides': 0.0}
    for k in sorted(losses):
        rel = abs(errs[k] - losses[k]) / max(abs(losses[k]), 1e-12)
        report.append('%-16s gpu %.7g oracle %.7g rel %.2e' % (k, errs[k], losses[k], rel))
        assert rel <= LOSS_RTOL, report[-1]
        assert abs(losses[k]) > 1e-4, report[-1]                       # a vanishing term would make the check vacuous
    for k in GRAD_KEYS:
        scale = float(g_ref[k].abs().max())
        err = float((g_gpu[k] - g_ref[k]).abs().max())
        report.append('%-44s max|g| %.3e  err/max %.2e' % (k, scale, err / max(scale, 1e-30)))
        assert scale > 1e-7, report[-1]
        worst['G'] = max(worst['G'], err / scale)
        assert err <= GRAD_RTOL * scale, report[-1]
    for k in sorted(d_ref):
        scale = float(d_ref[k].abs().max())
        err = float((d_gpu[k] - d_ref[k]).abs().max())
        own = float((d_gpu[k] - d_own[k]).abs().max())
        report.append('%-44s max|g| %.3e  err/max %.2e   (oracle deciding the kink sides itself: %.2e)' %
                      ('D.' + k, scale, err / max(scale, 1e-30), own / max(scale, 1e-30)))
        assert scale > 1e-7, report[-1]
        worst['D'], worst['D own sides'] = max(worst['D'], err / scale), max(worst['D own sides'], own / scale)
        assert err <= D_GRAD_RTOL * scale, report[-1]
    report.append('worst err/max: generator %.2e (bound %.0e)  discriminator %.2e (bound %.0e; %.2e with the oracle deciding the kink sides itself)'
                  % (worst['G'], GRAD_RTOL, worst['D'], D_GRAD_RTOL, worst['D own sides']))
    # the discriminator's weights were renormalised 3 x 13 times in place on both sides: same end state
    for k, v in env.discriminator.state_dict().items():
        ref = disc.sd[k]
        assert float((v.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max()), k
    print('\n'.join(report))
    if profile:
        return errs, prof
    return errs


def test_training_step_cfg3_geometry_reduced_width(tmp_path, monkeypatch):
    _step_and_compare(tmp_path, monkeypatch, gf_dim=8, kf_dim=4, df_dim=8, B=2)


def test_training_step_full_width(tmp_path, monkeypatch):
    """TAI_gray as configs[2] trains it (gf_dim 64, kf_dim 32, df_dim 64), B = 2."""
    _step_and_compare(tmp_path, monkeypatch, gf_dim=64, kf_dim=32, df_dim=64, B=2)


def test_training_step_full_width_at_the_32_clip_dispatch(tmp_path, monkeypatch, dispatch_at_32_clips):
    """The same full-width B = 2 step with every convolution routed as at configs[2]'s 32 clips per GPU (tests/conftest.py:
    DispatchAt): the ConvLSTM's 512 -> 1024 at 16 x 16, CombLayers, ContentEnc / DecCnn's 256-channel layers at 32 x 32, Residual 3 and
    the kernel network's 8 x 8 / 4 x 4 layers run _WinoConv3x3 / _WinoConv3x3Parts / tai_conv3x3_wino_wrw where the default thresholds
    send a 2-clip batch to MIOpen (conv_ops.WINO_MIN_WORKGROUPS).  Same oracle, same bounds; and no ATen convolution ran at all
    (environments.py:348-379, mcnet.py:259-294, SNDiscriminator.py:113-133)."""
    from conftest import miopen_convolutions
    _, prof = _step_and_compare(tmp_path, monkeypatch, gf_dim=64, kf_dim=32, df_dim=64, B=2, profile=True, tag='32-clip dispatch')
    # since round 5 that includes the discriminator (its 4 x 4 stride-2 layers as 3 x 3 layers on space-to-depth planes) and the weight
    # gradients of the 8 x 8 / 4 x 4 layers (rows widened to the weight-gradient kernel's 16 pixels): no ATen convolution in the update
    convs = miopen_convolutions(prof)
    assert not convs, sorted(set((c[0], tuple(c[1][0])) for c in convs))
    r = dispatch_at_32_clips.routes
    print('routes taken: %s' % r)
    assert r['wino'] > 100


def test_forward_train_of_32_clips_equals_the_2_clip_forward_at_its_dispatch(tmp_path, monkeypatch):
    """Clips are independent (no batch statistics anywhere in the generator): the training-mode forward of clips [0:2] inside a
    32-clip batch equals the forward of those two clips alone once the 2-clip batch takes the 32-clip batch's kernel routes -- which
    ties the oracle comparison above (B = 2, forced routes) to the batch configs[2] trains on."""
    from conftest import DispatchAt
    K = T = F = 5
    env, _ = _seeded_env(tmp_path, 64, 32, 64, K, T, F, 128, 128)
    clips = torch.from_numpy(synthetic.make_clips(32, K + T + F, 1, 128, 128, synthetic.SEEDS['cfg3']))
    P, GT, Fo = synthetic.split_clip(clips, K, T, F)
    env.K, env.T, env.F = K, T, F
    env.train()
    env.set_train_inputs(P, Fo, GT)
    env.forward_train()
    big = {k: v[:2].detach().clone() for k, v in env.gen_output.items()}
    env.gen_output = None
    DispatchAt(monkeypatch, 16)
    env.set_train_inputs(P[:2], Fo[:2], GT[:2])
    env.forward_train()
    for k, v in env.gen_output.items():
        scale = float(big[k].abs().max())
        err = float((v.detach() - big[k]).abs().max())
        print('%-24s max %.3e  err/max %.2e' % (k, scale, err / scale))
        assert scale > 0.05 and err <= 1e-5 * scale, (k, err, scale)


def test_training_step_short_context_nonsquare(tmp_path, monkeypatch):
    """sample_KTF draws K, F >= 2 and T >= 1 per step (environments.py:417-427): K != F (no direction fusion), T = 2."""
    _step_and_compare(tmp_path, monkeypatch, gf_dim=8, kf_dim=4, df_dim=8, B=2, K=4, T=2, F=3, H=64, W=96)


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
def test_graphed_update_matches_the_eager_update(tmp_path, monkeypatch):
    """train_step with graph_step (two eager updates, then one captured update replayed) against the same update run
    eagerly.  The comparison is made where it is well-posed: on the GRADIENTS of one update that both environments start
    from bit-identical state (weights, Adam moments and step counter, spectral-norm u vectors copied in place from the
    eager environment into the captured one's static tensors), before Adam turns rounding-level gradient elements into
    +-lr steps.  Every 3x3 layer is forced onto the in-tree Winograd kernels (fixed summation order) so the generator's
    gradients are reproducible; the discriminator's 4x4 stride-2 layers stay on MIOpen (atomics in its weight gradient),
    hence its own, looser bound.  Then: every parameter moved in that update, by at most Adam's bound."""
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO_MIN_WORKGROUPS', 0)
    K = T = F = 3
    H = W = 64
    B = 2
    LR = 1e-3
    clips = torch.from_numpy(synthetic.make_clips(4 * B, K + T + F, 1, H, W, synthetic.SEEDS['cfg3']))
    envs = {}
    for mode in ('eager', 'graph'):
        model = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16)
        env = TAITrainingEnvironment(model, str(tmp_path / mode), 'exp', [H, W], 1, ALPHA, BETA, LR, 0.5, 16, IP, DISC_T, K, T, F,
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
        for step in range(3):                          # graph: two eager updates, then capture + first replay
            P, GT, Fo = synthetic.split_clip(clips[step * B:(step + 1) * B], K, T, F)
            env.train_step(P, Fo, GT)
        envs[mode] = env
    eager, graph = envs['eager'], envs['graph']
    state = graph._step_graphs
    assert len(state) == 1 and 'graph' in next(iter(state.values()))
    assert all('graph' not in v for v in eager._step_graphs.values())

    # --- identical starting state for update 4, written IN PLACE into the tensors the captured graph reads
    with torch.no_grad():
        for part in ('generator', 'discriminator'):
            src, dst = getattr(eager, part), getattr(graph, part)
            for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
                b.copy_(a)
            for (name, ms), (_, md) in zip(src.named_modules(), dst.named_modules()):
                if hasattr(ms, 'Ip'):
                    md.u.copy_(ms.u)
        for o_src, o_dst in ((eager.optimizer_G, graph.optimizer_G), (eager.optimizer_D, graph.optimizer_D)):
            for ps, pd in zip(o_src.param_groups[0]['params'], o_dst.param_groups[0]['params']):
                ss, sd = o_src.state.get(ps, {}), o_dst.state.get(pd, {})
                assert set(ss) == set(sd)
                for key, v in ss.items():
                    sd[key].copy_(v)
    conv_ops.invalidate_derived()
    before = {part: {k: v.detach().clone() for k, v in getattr(graph, part).named_parameters()} for part in ('generator', 'discriminator')}

    P, GT, Fo = synthetic.split_clip(clips[3 * B:4 * B], K, T, F)
    eager.train_step(P, Fo, GT)
    graph.train_step(P, Fo, GT)                         # a replay
    torch.cuda.synchronize()

    # --- loss terms of that update
    e_err, g_err = eager.get_current_errors(), graph.get_current_errors()
    for k, v in e_err.items():
        assert abs(g_err[k] - v) <= 2e-5 * max(abs(v), 1e-3), (k, g_err[k], v)
    # --- gradients before Adam (p.grad still holds them after the step)
    report = []
    # measured on the GPU box: <= 9.2e-7 of a tensor's maximum (generator), <= 1.8e-6 (discriminator, MIOpen's atomics)
    for part, tol in (('generator', 2e-5), ('discriminator', 1e-4)):
        for (k, pe), (_, pg) in zip(getattr(eager, part).named_parameters(), getattr(graph, part).named_parameters()):
            if pe.grad is None:
                assert pg.grad is None and k.startswith('merge_residual1'), k       # never evaluated (tai.py:224-226)
                continue
            scale = float(pe.grad.abs().max())
            err = float((pg.grad - pe.grad).abs().max())
            report.append('%-14s %-44s max|g| %.3e err/max %.2e' % (part, k, scale, err / max(scale, 1e-30)))
            assert scale > 0, report[-1]
            assert err <= tol * scale, report[-1]
    print('\n'.join(report))
    # --- every parameter moved, by no more than Adam can move it in one update (lr x bias-corrected ratio <~ a few lr)
    for part in ('generator', 'discriminator'):
        for k, p in getattr(graph, part).named_parameters():
            if p.grad is None:
                continue
            delta = (p.detach() - before[part][k]).abs()
            if part == 'discriminator' and k.endswith('weight'):
                continue        # also renormalised in place by 3 x 7 spectral-norm passes: checked against the eager side below
            assert float(delta.max()) > 0.1 * LR, (part, k, float(delta.max()))
            assert float(delta.max()) <= 5 * LR, (part, k, float(delta.max()))
        for (k, pe), (_, pg) in zip(getattr(eager, part).named_parameters(), getattr(graph, part).named_parameters()):
            # same start, same gradients to rounding: the updated weights agree except where a gradient element is at
            # rounding level (Adam normalises each element by its own magnitude), so the MEAN difference is tiny
            assert float((pe.detach() - pg.detach()).abs().mean()) <= 2e-5, (part, k)
    # after replays an eager forward sees the CURRENT weights (derived Winograd filters rebuilt)
    P, GT, Fo = synthetic.split_clip(clips[:B], K, T, F)
    outs = []
    for env in (eager, graph):
        env.eval()
        with torch.no_grad():
            outs.append(env.generator(T, P.to(DEV), Fo.to(DEV))['pred'])
    assert torch.isfinite(outs[1]).all()
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-3


@pytest.mark.gpu
def test_captured_workspace_survives_a_larger_eager_request():
    """conv_ops keeps one growing scratch buffer per device for the weight-gradient kernels; a captured graph bakes in its
    pointer.  A later eager call with a larger request used to REPLACE the buffer and hand the old block back to the
    allocator while replays still wrote partial sums into it: superseded buffers are now kept alive once a
    capture has seen them."""
    from video_frame_inpainting_amd import conv_ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4, 64, 32, 32, generator=g).to(DEV)
    gy = torch.randn(4, 64, 32, 32, generator=g).to(DEV)
    conv_ops._WRW_WORKSPACE.current.pop(x.device, None)         # start from no workspace on this device
    want = conv_ops.wino_weight_grad(x, gy)
    assert want is not None
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        got = conv_ops.wino_weight_grad(x, gy)
    small_ws = conv_ops._WRW_WORKSPACE.current[x.device]
    xb = torch.randn(32, 64, 64, 64, generator=g).to(DEV)
    gb = torch.randn(32, 64, 64, 64, generator=g).to(DEV)
    conv_ops.wino_weight_grad(xb, gb)                            # larger request: a new buffer ...
    assert conv_ops._WRW_WORKSPACE.current[x.device].data_ptr() != small_ws.data_ptr()
    assert any(t.data_ptr() == small_ws.data_ptr() for t in conv_ops._WRW_WORKSPACE.retired)     # ... the old one retired, not freed
    del small_ws
    junk = [torch.full((1 << 20,), float('nan'), device=DEV) for _ in range(64)]      # whatever the allocator hands out now
    graph.replay()
    for t in junk:
        t.fill_(float('nan'))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert all(bool(torch.isnan(t).all()) for t in junk)         # the replay wrote into nobody else's memory


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


def _force_in_tree_routes(monkeypatch):
    """Small reference-run shapes through the routes of the training batch: every 3x3 form the in-tree Winograd kernels can take, they take
    (the discriminator's 4x4 stride-2 layers as 3x3 layers on space-to-depth planes: sn_discriminator._s2d_applies)."""
    from video_frame_inpainting_amd import conv_ops
    monkeypatch.setattr(conv_ops, 'WINO_MIN_WORKGROUPS', 1)
    monkeypatch.setattr(conv_ops, 'WINO43_MIN_WORKGROUPS', 1)


@pytest.mark.gpu
@pytest.mark.parametrize('in_tree', [False, True])
@pytest.mark.parametrize('tag, c_dim', [('disc_gray', 1), ('disc_color', 3)])
def test_gpu_discriminator_matches_reference_run_over_two_calls(golden_dir, tag, c_dim, in_tree, monkeypatch):
    """The product's all-windows-in-one-pass discriminator (renormalisations up front, one factor per window) against
    the reference's window-by-window SNDiscriminator.forward (:140-159) over two consecutive calls: logits, the
    in-place renormalised weights and the persistent u vectors.  ``in_tree``: the convolutions on the Winograd kernels (space-to-depth
    form) as at the training batch, instead of MIOpen's, which these 32 x 32 shapes take by default."""
    import os
    from video_frame_inpainting_amd.sn_discriminator import SNDiscriminator
    if in_tree:
        _force_in_tree_routes(monkeypatch)
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
@pytest.mark.parametrize('in_tree', [False, True])
def test_gpu_discriminator_gradients_match_reference_run_single_window(golden_dir, in_tree, monkeypatch):
    import os
    import torch.nn.functional as Fn
    from video_frame_inpainting_amd.sn_discriminator import SNDiscriminator
    if in_tree:
        _force_in_tree_routes(monkeypatch)
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


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(13, 2, 64, 128, 64), (13, 2, 3, 64, 128), (13, 32, 128, 256, 32), (13, 32, 256, 512, 16)])
def test_discriminator_layer_on_space_to_depth_planes_matches_the_direct_form(shape, monkeypatch):
    """One 4x4 stride-2 layer + LeakyReLU of the sliding-window discriminator (SNDiscriminator.py:113-133) as the 3x3 layer over
    space-to-depth planes on the in-tree Winograd kernels (forward, input gradient, weight and bias gradient) against float64 direct
    convolutions of the same operands, next to the MIOpen route the layer took until round 5; the in-tree gradients are bit-reproducible."""
    import torch.nn.functional as Fn
    from conftest import miopen_convolutions
    from video_frame_inpainting_amd import conv_ops, sn_discriminator as snd
    nw, B, C, K, H = shape
    g = torch.Generator().manual_seed(C + K + H)
    x = torch.randn(nw * B, C, H, H, generator=g).to(DEV)
    w = (torch.randn(K, C, 4, 4, generator=g) * (2.0 / (16 * C)) ** 0.5).to(DEV)
    b = (0.1 * torch.randn(K, generator=g)).to(DEV)
    inv = (0.5 + torch.rand(nw, generator=g)).to(DEV)
    gy = torch.randn(nw * B, K, H // 2, H // 2, generator=g).to(DEV)
    if B == 2:
        monkeypatch.setattr(conv_ops, 'WINO_MIN_WORKGROUPS', 1)
        monkeypatch.setattr(conv_ops, 'WINO43_MIN_WORKGROUPS', 1)

    def run(route):
        with monkeypatch.context() as mp:
            if route == 'direct':
                mp.setattr(snd, '_s2d_applies', lambda *a: False)
            xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU], record_shapes=True) as prof:
                y = snd._WindowScaledConvLReLU.apply(xr, wr, wr.detach(), br, inv, nw, (2, 2), (1, 1), 0.2)
                y.backward(gy)
            return (y.detach(), xr.grad, wr.grad, br.grad), miopen_convolutions(prof)

    got, convs = run('s2d')
    assert not convs, convs                          # every convolution of the layer ran in-tree (the 8 x 8 layer's weight gradient on widened rows)
    again, _ = run('s2d')
    assert all(torch.equal(a, c) for a, c in zip(got, again))
    old, convs_old = run('direct')
    assert convs_old
    xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    z = Fn.conv2d(xd, wd, None, 2, 1)
    z = (z.view(nw, B, -1) * inv.double().view(nw, 1, 1)).view_as(z) + bd.view(1, -1, 1, 1)
    yd = Fn.leaky_relu(z, 0.2)
    # the fp32 routes may put a pre-activation within rounding of 0 on the other side of the kink: differentiate the float64 form with
    # the product's sides (its gradient mask) so that the comparison is about the convolutions
    side = (got[0] > 0).double()
    slope = side + 0.2 * (1 - side)
    (z * slope).backward(gy.double())
    # the weight gradient takes the pre-activation gradient WITHOUT the window's factor: every window differentiates with respect to its own
    # weight tensor w0 * inv_scale[t] and the results accumulate in the parameter (_WindowScaledConv's docstring; SNDiscriminator.py:60-68)
    wu = w.double().requires_grad_()
    Fn.conv2d(x.double(), wu, None, 2, 1).backward(gy.double() * slope)
    for name, a, o, r in zip(('y', 'gx', 'gw', 'gb'), got, old, (yd.detach(), xd.grad, wu.grad, bd.grad)):
        scale = float(r.abs().max())
        e_new, e_old = float((a.double() - r).abs().max()) / scale, float((o.double() - r).abs().max()) / scale
        # (the MIOpen route's gradients are scored here with the IN-TREE route's kink sides, so its own rounding flips count against it:
        # only its y is comparable)
        print('%s %s: in-tree %.2e of the maximum%s' % (shape, name, e_new, '  (MIOpen route: %.2e)' % e_old if name == 'y' else ''))
        assert e_new <= 2e-5, (name, e_new, e_old)
