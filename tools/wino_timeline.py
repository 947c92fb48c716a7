"""Per-workgroup phase times of the Winograd-MFMA convolution from its shader-clock stamps, for both workgroup shapes
(64 channels x 64 tiles; 128 x 32 when K is a multiple of 128).  --skip: also the timing-only ablations."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from video_frame_inpainting_amd import _native
L = _native.lib()
shapes = [(64, 256, 128, 64, 64)] if '--skip' in sys.argv else [(64, 64, 64, 128, 128), (64, 256, 128, 64, 64), (64, 512, 1024, 16, 16), (32, 51, 51, 128, 128)]
levels = [0, 1, 4, 2, 5] if '--skip' in sys.argv else [0]
for level in levels:
    L.tai_conv3x3_wino_timeline_skip(level)
    print('skip level', level, '(0 full kernel; timing-only ablations, results wrong: 1 no patch transform, 4 + no patch loads, 2 + no V writes, 5 + no output stores)')
    for (N, C, K, H, W) in shapes:
        for tall in ((0, 1) if K % 128 == 0 else (0,)):
            L.tai_conv3x3_wino_set_tall(tall)
            x = torch.randn(N, C, H, W, device='cuda'); w = torch.randn(K, C, 3, 3, device='cuda') * .05; b = torch.zeros(K, device='cuda')
            U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda'); y = torch.empty(N, K, H, W, device='cuda')
            s = torch.cuda.current_stream().cuda_stream
            _native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, s), 'tw')
            wgs = ((N * H * W // 4 + 63) // 64) * ((K + 63) // 64)          # the same count for both shapes
            st = torch.zeros(wgs * 64, dtype=torch.int64, device='cuda')
            for _ in range(3):
                _native.check(L.tai_conv3x3_wino_forward_timeline(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, st.data_ptr(), s), 'fw')
            torch.cuda.synchronize()
            t = st.cpu().numpy().reshape(wgs, 64).astype(np.float64)
            nch = min((C + 7) // 8, 26)
            pro = t[:, 1] - t[:, 0]; loop = t[:, 2] - t[:, 1]; epi = t[:, 3] - t[:, 2]; tot = t[:, 3] - t[:, 0]
            ch = np.diff(np.concatenate([t[:, 1:2], t[:, 4:4 + nch]], axis=1), axis=1)
            ghz = np.median(tot / np.maximum(t[:, 61] - t[:, 60], 1)) * 0.1      # shader cycles per 100 MHz tick
            print('x(%d,%d,%d,%d)->%d [%s]: %d workgroups, %d chunks; clocks: prologue %.0f  loop %.0f (%.0f/chunk; ideal 4096)  epilogue %.0f  total %.0f  shader clock %.2f GHz'
                  % (N, C, H, W, K, '128 x 32' if tall else '64 x 64', wgs, (C + 7) // 8, np.median(pro), np.median(loop), np.median(loop) / max((C + 7) // 8, 1), np.median(epi), np.median(tot), ghz))
            sub = [np.median(t[:, 40] - t[:, 0]), np.median(t[:, 41] - t[:, 40]), np.median(t[:, 42] - t[:, 41]), np.median(t[:, 43] - t[:, 42]),
                   np.median(t[:, 44] - t[:, 43]), np.median(t[:, 1] - t[:, 44]), np.median(t[:, 45] - t[:, 2]), np.median(t[:, 3] - t[:, 45])]
            print('   prologue: set-up %.0f | loads + DMA issued %.0f | accumulator init %.0f | chunk 0 landed + transformed %.0f | LDS writes %.0f | waits + barrier %.0f'
                  '     epilogue: inverse transform %.0f | activation + stores %.0f' % tuple(sub))
            print('   per-chunk median by index:', ' '.join('%.0f' % v for v in np.median(ch, axis=0)[:16]), ' p90 of all chunks %.0f' % np.percentile(ch, 90), flush=True)
L.tai_conv3x3_wino_timeline_skip(0)
L.tai_conv3x3_wino_set_tall(1)
