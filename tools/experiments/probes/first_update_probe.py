#!/usr/bin/env python3
"""the first training update took 20 s -- 19 s of GPU time in MIOpen's find step benchmarking every solver
(naive_conv_* reference kernels at 230-260 ms each) for the discriminator's 4x4 stride-2 convolutions, per rank, at every start.
This probe runs the first two updates and three timed ones in a FRESH process per MIOPEN_FIND_MODE value and prints first-update wall
time and steady-state ms per update (a find mode that skips the search may also pick slower kernels: both numbers matter).
Usage: python tools/first_update_probe.py            (parent: one child per mode)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ['(unset)', 'NORMAL', 'FAST', 'HYBRID', 'DYNAMIC_HYBRID']


def child():
    sys.path.insert(0, ROOT)
    import contextlib
    import tempfile
    import torch
    import video_frame_inpainting_amd as vfi
    from video_frame_inpainting_amd import synthetic
    from video_frame_inpainting_amd.environments import create_training_environment
    dev = torch.device('cuda:0')
    B, K, T, F, H, W = 32, 5, 5, 5, 128, 128
    with contextlib.redirect_stdout(sys.stderr):
        env = create_training_environment(vfi.create_model('TAI_gray'), 1, tempfile.mkdtemp(prefix='tai_probe_'), 'probe', 5, 5, 5, [H, W],
                                          1.0, 0.02, 1e-4, 0.5, 64, 3, 3, [0, 0], device=dev)
    env.sync_replicas()
    clips = torch.from_numpy(synthetic.make_clips(B, K + T + F, 1, H, W, synthetic.SEEDS['cfg3']))
    P, GT, Fo = synthetic.split_clip(clips, K, T, F)

    def step():
        env.K, env.T, env.F = K, T, F
        env.train()
        env.train_step(P, Fo, GT)
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); first = time.perf_counter() - t0
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); second = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t0) / 3
    errs = env.get_current_errors()
    print('RESULT mode=%s first=%.2f s second=%.3f s steady=%.1f ms loss_G=%.6f' % (
        os.environ.get('MIOPEN_FIND_MODE', '(unset)'), first, second, steady * 1e3, errs.get('G_loss', float('nan'))), flush=True)


if __name__ == '__main__':
    if os.environ.get('TAI_PROBE_CHILD') == '1':
        child()
    else:
        for mode in (sys.argv[1].split(',') if len(sys.argv) > 1 else MODES):
            env = dict(os.environ, TAI_PROBE_CHILD='1')
            env.pop('MIOPEN_FIND_MODE', None)
            if mode != '(unset)':
                env['MIOPEN_FIND_MODE'] = mode
            # a private, empty user db per child: every mode starts from the same (cold) state
            import tempfile
            # TAI_PROBE_DB=<dir>: every child shares that user find-db (first child fills it, the others start from it)
            env['MIOPEN_USER_DB_PATH'] = os.environ.get('TAI_PROBE_DB') or tempfile.mkdtemp(prefix='miopen_db_')
            os.makedirs(env['MIOPEN_USER_DB_PATH'], exist_ok=True)
            env['MIOPEN_CUSTOM_CACHE_DIR'] = tempfile.mkdtemp(prefix='miopen_cache_')
            t0 = time.time()
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
            print('%-16s %s   (process %.0f s, rc %d)' % (mode, line[0] if line else 'no result: ' + r.stderr[-300:], time.time() - t0, r.returncode), flush=True)
