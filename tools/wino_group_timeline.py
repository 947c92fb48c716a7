"""Shader-clock time of each of the 16 MFMA groups in the first two chunks of the Winograd-MFMA convolution."""
import sys, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/tools') else os.getcwd())
import numpy as np, torch
from video_frame_inpainting_amd import _native
L = _native.lib()
N, C, K, H, W = 64, 64, 64, 128, 128
x = torch.randn(N, C, H, W, device='cuda'); w = torch.randn(K, C, 3, 3, device='cuda') * .05; b = torch.zeros(K, device='cuda')
U = torch.empty(L.tai_conv3x3_wino_weight_floats(K, C), device='cuda'); y = torch.empty(N, K, H, W, device='cuda')
_native.check(L.tai_conv3x3_wino_transform_weights(w.data_ptr(), U.data_ptr(), K, C, None), 'tw')
wgs = ((N * H * W // 4 + 63) // 64) * ((K + 63) // 64)
st = torch.zeros(wgs * 64, dtype=torch.int64, device='cuda')
for mode in (7,):
  L.tai_conv3x3_wino_timeline_skip(mode); print('per-group clocks (16 groups of 4 MFMAs = 256 clocks each when nothing else runs)')
  for _ in range(3):
      _native.check(L.tai_conv3x3_wino_forward_timeline(x.data_ptr(), U.data_ptr(), b.data_ptr(), y.data_ptr(), N, C, K, H, W, st.data_ptr(), None), 'fw')
  torch.cuda.synchronize()
  t = st.cpu().numpy().reshape(wgs, 64).astype(np.float64)
  for ch in range(2):
    g = t[:, 30 + 16 * ch: 46 + 16 * ch]
    start = t[:, 1] if ch == 0 else t[:, 4]
    d = np.diff(np.concatenate([start[:, None], g], axis=1), axis=1)
    print('chunk', ch, 'per-group median clocks:', ' '.join('%.0f' % v for v in np.median(d, axis=0)))

print('chunk 0 group 0: prologue barrier -> first MFMA issued %.0f clocks' % np.median(t[:, 62] - t[:, 1]))
