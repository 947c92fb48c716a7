#!/usr/bin/env python3
"""Which part of a training update survives hipGraph capture: runs stages in child processes (a fault in the HIP runtime
kills the child only).  python tools/graph_step_bisect.py [stage]"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAGES = ['forward', 'loss_G', 'backward_G', 'step_G', 'loss_D', 'backward_D', 'step_D']
EXTRA = ['bwd_rec_only', 'bwd_gan_only', 'bwd_gan_detached_gen', 'bwd_two_graphs']
if len(sys.argv) == 1:
    for st in (os.environ['EXTRA'].split(',') if os.environ.get('EXTRA') else STAGES):
        r = subprocess.run([sys.executable, '-X', 'faulthandler', __file__, st], capture_output=True, text=True)
        tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
        print('%-12s rc=%d  %s' % (st, r.returncode, ' | '.join(tail)), flush=True)
    sys.exit(0)
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.environments import TAITrainingEnvironment
stage = STAGES.index(sys.argv[1]) if sys.argv[1] in STAGES else -1
extra = sys.argv[1]
dev = torch.device('cuda:0')
K = T = F = 3; H = W = 64; B = 2
model = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16)
env = TAITrainingEnvironment(model, '/tmp/ckpt_bisect', 'x', [H, W], 1, 1.0, 0.02, 1e-3, 0.5, 16, 3, 3, K, T, F, [0, 0], device=dev, graph_step=True)
clips = torch.from_numpy(synthetic.make_clips(B, K + T + F, 1, H, W, 1003))
P, GT, Fo = synthetic.split_clip(clips, K, T, F)
env.K, env.T, env.F = K, T, F
env.train()
WARM = os.environ.get('WARM', 'full')
for _ in range(2):
    env.set_train_inputs(P, Fo, GT); env.forward_train()
    if WARM == 'full':
        env.optimize_parameters()
    elif WARM == 'nostep':
        env.optimizer_G.zero_grad(); env.compute_loss_G(); env.loss_G.backward()
        env.optimizer_D.zero_grad(); env.compute_loss_D(); env.loss_D.backward()
    elif WARM == 'gstep':
        env.optimizer_G.zero_grad(); env.compute_loss_G(); env.loss_G.backward(); env.optimizer_G.step()
    elif WARM == 'recstep':
        env.optimizer_G.zero_grad()
        gt = env._time_major_01(env.gt_middle_frames); out = env._time_major_01(env.gen_output['pred'])
        (env.loss_Lp(out, gt) + env.loss_gdl(out, gt)).backward(); env.optimizer_G.step()
    elif WARM == 'meanonly':
        env.generator.zero_grad(); env.gen_output['pred'].mean().backward()
    elif WARM == 'mean3':
        env.generator.zero_grad(); out = env.gen_output; (out['pred'].mean() + out['pred_forward'].mean() + out['pred_backward'].mean()).backward()
    elif WARM == 'recnostep':
        env.optimizer_G.zero_grad()
        gt = env._time_major_01(env.gt_middle_frames); out = env._time_major_01(env.gen_output['pred'])
        (env.loss_Lp(out, gt) + env.loss_gdl(out, gt)).backward()
env._prepare_capture()
torch.cuda.synchronize()
def detach_all():
    for k in env._STEP_OUTPUTS:
        v = getattr(env, k, None)
        if isinstance(v, dict):
            setattr(env, k, {name: t.detach() for name, t in v.items()})
        elif torch.is_tensor(v):
            setattr(env, k, v.detach())

if os.environ.get('PRE_DETACH', '1') == '1':
    detach_all()
g = torch.cuda.CUDAGraph()
if stage < 0:
    import torch.nn.functional as Fn
    if extra in ('m0', 'm1', 'm2', 'm3'):
        with torch.cuda.graph(g):
            env.forward_train()
            if extra in ('m1', 'm2'):
                env.optimizer_G.zero_grad()
            if extra == 'm3':
                env.generator.zero_grad()
            if extra == 'm2':
                gt = env._time_major_01(env.gt_middle_frames); out = env._time_major_01(env.gen_output['pred'])
            env.gen_output['pred'].mean().backward()
            detach_all()
        print('captured'); g.replay(); torch.cuda.synchronize(); print('replayed ok')
        sys.exit(0)
    with torch.cuda.graph(g):
        env.forward_train()
        env.optimizer_G.zero_grad()
        gt = env._time_major_01(env.gt_middle_frames); out = env._time_major_01(env.gen_output['pred'])
        rec = env.loss_Lp(out, gt) + env.loss_gdl(out, gt)
        if extra == 'bwd_rec_only':
            rec.backward()
        elif extra == 'bwd_mean_only':
            env.gen_output['pred'].mean().backward()
        elif extra == 'bwd_mse_only':
            env.loss_Lp(out, gt).backward()
        elif extra == 'bwd_gdl_only':
            env.loss_gdl(out, gt).backward()
        elif extra == 'bwd_mse_plain':
            Fn.mse_loss(env.gen_output['pred'], env.gt_middle_frames).backward()
        else:
            pred = env.gen_output['pred'].detach().requires_grad_(True) if extra == 'bwd_gan_detached_gen' else env.gen_output['pred']
            fake = torch.cat([env.preceding_frames, pred, env.following_frames], dim=1)
            h = env.discriminator(fake)
            gan = env.loss_d(h, torch.ones_like(h))
            if extra == 'bwd_two_graphs':
                (rec + 0.02 * gan).backward()
            else:
                gan.backward()
        rec = gan = h = fake = pred = out = gt = None
        detach_all()
    print('captured'); g.replay(); torch.cuda.synchronize(); print('replayed ok')
    sys.exit(0)
with torch.cuda.graph(g):
    env.forward_train()
    if stage >= 1:
        env.optimizer_G.zero_grad(); env.compute_loss_G()
    if stage >= 2:
        env.loss_G.backward()
    if stage >= 3:
        env.optimizer_G.step()
    if stage >= 4:
        env.optimizer_D.zero_grad(); env.compute_loss_D()
    if stage >= 5:
        env.loss_D.backward()
    if stage >= 6:
        env.optimizer_D.step()
    for k in env._STEP_OUTPUTS:
        v = getattr(env, k, None)
        if isinstance(v, dict):
            setattr(env, k, {name: t.detach() for name, t in v.items()})
        elif torch.is_tensor(v):
            setattr(env, k, v.detach())
    v = None
print('captured'); g.replay(); torch.cuda.synchronize(); print('replayed ok')
