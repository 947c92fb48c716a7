#!/usr/bin/env python3
"""After one eager update with the step outputs detached: which tensors with autograd history are still alive?"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import synthetic
from video_frame_inpainting_amd.environments import TAITrainingEnvironment
dev = torch.device('cuda:0')
K = T = F = 3; H = W = 64; B = 2
model = vfi.TAIFillInModel(16, 1, 3, 51, num_block=5, kf_dim=16)
env = TAITrainingEnvironment(model, '/tmp/ckpt_bisect', 'x', [H, W], 1, 1.0, 0.02, 1e-3, 0.5, 16, 3, 3, K, T, F, [0, 0], device=dev, graph_step=True)
clips = torch.from_numpy(synthetic.make_clips(B, K + T + F, 1, H, W, 1003))
P, GT, Fo = synthetic.split_clip(clips, K, T, F)
env.K, env.T, env.F = K, T, F
env.train()
env.set_train_inputs(P, Fo, GT); env.forward_train(); env.optimize_parameters()
for k in env._STEP_OUTPUTS:
    v = getattr(env, k, None)
    if isinstance(v, dict):
        setattr(env, k, {name: t.detach() for name, t in v.items()})
    elif torch.is_tensor(v):
        setattr(env, k, v.detach())
del v
n = 0
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.grad_fn is not None:
            n += 1
            refs = [type(r).__name__ for r in gc.get_referrers(o)][:4]
            print('live with history:', tuple(o.shape), type(o.grad_fn).__name__, refs)
    except Exception:
        pass
print('count', n)
gc.collect()
n2 = sum(1 for o in gc.get_objects() if torch.is_tensor(o) and o.grad_fn is not None)
print('after gc.collect()', n2)
