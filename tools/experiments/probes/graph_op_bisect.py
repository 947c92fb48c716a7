#!/usr/bin/env python3
"""Which operator's backward survives hipGraph capture (child process per case)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ['gen_keep_out', 'gen_keep_out_detached', 'gen_pred_only', 'gen_zero_after', 'gen_train_T', 'wino_fresh', 'wino_fresh_fwd', 'gen_fresh', 'sepconv', 'sepconv_pad', 'gdl', 'disc', 'gen_small', 'wino3x3', 'wino3x3_small', 'kxk', 'miopen_conv', 'maxpool', 'upsample', 'thin_in', 'convT', 'sn_conv', 'lstm', 'cat_split']
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, c], capture_output=True, text=True)
        out = (r.stdout + r.stderr)
        print('%-14s rc=%d %s' % (c, r.returncode, 'ok' if 'replayed ok' in out else out.strip().splitlines()[-1][:150] if out.strip() else ''), flush=True)
    sys.exit(0)
import torch, torch.nn.functional as F
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import conv_ops
case = sys.argv[1]
dev = 'cuda:0'
torch.manual_seed(0)

def run(fn, between=None):
    for _ in range(2):
        fn()
    if between:
        between()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    print('captured', flush=True); g.replay(); torch.cuda.synchronize(); print('replayed ok')

if case in ('wino_fresh', 'wino_fresh_fwd'):
    x = torch.randn(8, 64, 64, 64, device=dev, requires_grad=True); w = torch.randn(64, 64, 3, 3, device=dev, requires_grad=True); b = torch.randn(64, device=dev, requires_grad=True)
    def bump():
        with torch.no_grad():
            w.add_(0.001)
    if case == 'wino_fresh':
        run(lambda: conv_ops.conv_bias_act(x, w, b, 1, 'relu').sum().backward(), bump)
    else:
        def f():
            with torch.no_grad():
                conv_ops.conv_bias_act(x, w, b, 1, 'relu')
        run(f, bump)
elif case in ('gen_keep_out', 'gen_keep_out_detached'):
    m = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16).to(dev)
    P = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1; Fo = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1
    keep = {}
    def f():
        out = m(3, P, Fo); m.zero_grad(); out['pred'].mean().backward()
        keep['out'] = {k: v.detach() for k, v in out.items()} if case == 'gen_keep_out_detached' else out
    run(f)
elif case in ('gen_pred_only', 'gen_zero_after', 'gen_train_T'):
    m = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16).to(dev)
    P = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1; Fo = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1
    def f():
        if case == 'gen_pred_only':
            m.zero_grad(); out = m(3, P, Fo); out['pred'].mean().backward()
        elif case == 'gen_zero_after':
            out = m(3, P, Fo); m.zero_grad(); (out['pred'].mean() + out['pred_forward'].mean() + out['pred_backward'].mean()).backward()
        else:
            m.train(); m.zero_grad(); out = m(3, P, Fo); print(sorted(out.keys())); (out['pred'].mean() + out['pred_forward'].mean() + out['pred_backward'].mean()).backward()
    run(f)
elif case in ('gen_env', 'gen_init', 'gen_env_inputs'):
    from video_frame_inpainting_amd.environments import TAITrainingEnvironment
    from video_frame_inpainting_amd.util import weights_init
    from video_frame_inpainting_amd import synthetic
    m = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16)
    if case == 'gen_init':
        m = m.to(dev); m.apply(weights_init)
    else:
        env = TAITrainingEnvironment(m, '/tmp/ckpt_bisect', 'x', [64, 64], 1, 1.0, 0.02, 1e-3, 0.5, 16, 3, 3, 3, 3, 3, [0, 0], device=torch.device(dev), graph_step=True)
        m = env.generator
    if case == 'gen_env_inputs':
        clips = torch.from_numpy(synthetic.make_clips(2, 9, 1, 64, 64, 1003))
        P, GT, Fo = synthetic.split_clip(clips, 3, 3, 3)
        P, Fo = P.contiguous().to(dev), Fo.contiguous().to(dev)
    else:
        P = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1; Fo = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1
    def f():
        m.zero_grad(); out = m(3, P, Fo); (out['pred'].mean() + out['pred_forward'].mean() + out['pred_backward'].mean()).backward()
    run(f)
elif case == 'gen_fresh':
    m = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16).to(dev)
    P = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1; Fo = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1
    def f():
        m.zero_grad(); out = m(3, P, Fo); (out['pred'].mean() + out['pred_forward'].mean() + out['pred_backward'].mean()).backward()
    def bump():
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1e-4)
    run(f, bump)
elif case == 'sepconv':
    from video_frame_inpainting_amd.separable_convolution import SeparableConvolution
    x = torch.randn(4, 1, 114, 114, device=dev, requires_grad=True); v = torch.randn(4, 51, 64, 64, device=dev, requires_grad=True); h = torch.randn(4, 51, 64, 64, device=dev, requires_grad=True)
    run(lambda: SeparableConvolution.apply(x, v, h, 51).sum().backward())
elif case == 'sepconv_pad':
    from video_frame_inpainting_amd.separable_convolution import SeparableConvolution
    x = torch.randn(4, 1, 64, 64, device=dev, requires_grad=True); v = torch.randn(4, 51, 64, 64, device=dev, requires_grad=True); h = torch.randn(4, 51, 64, 64, device=dev, requires_grad=True)
    run(lambda: SeparableConvolution.apply(F.pad(x, (25, 25, 25, 25), mode='replicate'), v, h, 51).sum().backward())
elif case == 'gdl':
    from video_frame_inpainting_amd.losses import GDL
    x = torch.rand(6, 1, 64, 64, device=dev, requires_grad=True); y = torch.rand(6, 1, 64, 64, device=dev)
    l = GDL()
    run(lambda: (l(x, y) + F.mse_loss(x, y)).backward())
elif case == 'disc':
    from video_frame_inpainting_amd.sn_discriminator import SNDiscriminator
    d = SNDiscriminator((64, 64), 1, 3, 16, 3).to(dev); x = torch.randn(2, 9, 1, 64, 64, device=dev, requires_grad=True)
    run(lambda: F.binary_cross_entropy_with_logits(d(x), torch.ones(2, 7, device=dev)).backward())
elif case == 'gen_small':
    m = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16).to(dev)
    P = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1; Fo = torch.rand(2, 3, 1, 64, 64, device=dev) * 2 - 1
    def f():
        m.zero_grad(); out = m(3, P, Fo); (out['pred'].mean() + out['pred_forward'].mean() + out['pred_backward'].mean()).backward()
    run(f)
elif case in ('wino3x3', 'wino3x3_small'):
    N = 32 if case == 'wino3x3' else 2
    x = torch.randn(N, 64, 64, 64, device=dev, requires_grad=True); w = torch.randn(64, 64, 3, 3, device=dev, requires_grad=True); b = torch.randn(64, device=dev, requires_grad=True)
    def f():
        y = conv_ops.conv_bias_act(x, w, b, 1, 'relu'); print(type(y.grad_fn).__name__); y.sum().backward()
    run(f)
elif case == 'kxk':
    x = torch.randn(16, 64, 64, 64, device=dev, requires_grad=True); w = torch.randn(128, 64, 5, 5, device=dev, requires_grad=True); b = torch.randn(128, device=dev, requires_grad=True)
    def f():
        y = conv_ops.conv_bias_act(x, w, b, 2, 'relu'); print(type(y.grad_fn).__name__); y.sum().backward()
    run(f)
elif case == 'miopen_conv':
    x = torch.randn(4, 16, 64, 64, device=dev, requires_grad=True); w = torch.randn(16, 16, 4, 4, device=dev, requires_grad=True)
    run(lambda: F.conv2d(x, w, None, 2, 1).sum().backward())
elif case == 'maxpool':
    x = torch.randn(4, 16, 64, 64, device=dev, requires_grad=True)
    run(lambda: F.max_pool2d(x, 2).sum().backward())
elif case == 'upsample':
    x = torch.randn(4, 16, 32, 32, device=dev, requires_grad=True)
    run(lambda: F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=True).sum().backward())
elif case == 'thin_in':
    x = torch.randn(4, 1, 64, 64, device=dev, requires_grad=True); w = torch.randn(16, 1, 5, 5, device=dev, requires_grad=True); b = torch.randn(16, device=dev, requires_grad=True)
    run(lambda: conv_ops.conv_bias_act(x, w, b, 2, 'relu').sum().backward())
elif case == 'convT':
    x = torch.randn(4, 16, 64, 64, device=dev, requires_grad=True); w = torch.randn(16, 1, 3, 3, device=dev, requires_grad=True); b = torch.randn(1, device=dev, requires_grad=True)
    run(lambda: conv_ops.conv_bias_act(x, w, b, 1, 'tanh', transposed=True).sum().backward())
elif case == 'sn_conv':
    from video_frame_inpainting_amd.sn_discriminator import SNConv2d
    m = SNConv2d(3, 16, 4, 2, 1, Ip=3).to(dev); x = torch.randn(4, 3, 64, 64, device=dev, requires_grad=True)
    run(lambda: m(x).sum().backward())
elif case == 'lstm':
    from video_frame_inpainting_amd.mcnet import ConvLstmCell
    m = ConvLstmCell(3, 16, 16).to(dev) if hasattr(ConvLstmCell, '__init__') else None
    x = torch.randn(2, 16, 16, 16, device=dev, requires_grad=True); st = torch.zeros(2, 32, 16, 16, device=dev)
    run(lambda: m(x, st)[0].sum().backward())
elif case == 'cat_split':
    x = torch.randn(4, 16, 32, 32, device=dev, requires_grad=True)
    run(lambda: torch.cat([x, x * 2], 1)[:, 8:24].contiguous().sum().backward())
