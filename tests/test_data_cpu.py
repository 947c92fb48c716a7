"""Clip I/O (video_frame_inpainting_amd/data.py) against the reference's dataset semantics (src/data/base_dataset.py):
list formats, clip labels, the per-frame pipeline, and predict.py on a list of frame directories."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd import data as vdata
from video_frame_inpainting_amd.util import bgr2gray


def _frames(T=12, H=24, W=32, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    return np.stack([np.roll(base, t, axis=1) for t in range(T)])            # [T, H, W, 3] RGB, moving pattern


def _png_dir(root, name, frames):
    d = os.path.join(root, name)
    os.makedirs(d)
    for t, f in enumerate(frames):
        Image.fromarray(f).save(os.path.join(d, 'frame_%04d.png' % t))
    return d


def test_contiguous_list_labels_and_pipeline(tmp_path):
    fr = _frames()
    d = _png_dir(str(tmp_path), 'person01_walking', fr)
    lst = tmp_path / 'list.txt'
    lst.write_text('%s 3-9\n%s\n' % (d, d))
    ds = vdata.ContiguousVideoClipDataset(3, str(lst), 7, False, False, (24, 32), False, (8, 0))
    assert len(ds) == 2
    it = ds[0]
    assert it['clip_label'] == 'person01_walking_3-9'                         # basename_a-b, 1-indexed inclusive (:188)
    x = it['targets']
    assert x.shape == (7, 3, 24 + 8, 32) and x.dtype == torch.float32
    # the only 7-frame window of frames 3..9 (1-indexed) is frames 2..8 (0-indexed); BGR order; [-1, 1]; bottom pad = -1
    want = torch.from_numpy(fr[2:9, :, :, ::-1].copy()).permute(0, 3, 1, 2).float() / 255 * 2 - 1
    assert torch.allclose(x[:, :, :24], want, atol=1e-6)
    assert torch.all(x[:, :, 24:] == -1.0)
    assert ds[1]['clip_label'] == 'person01_walking_1-12'                     # whole video when no range is given


def test_gray_and_npy_source_and_disjoint_list(tmp_path):
    fr = _frames(T=15)
    np.save(tmp_path / 'clip.npy', fr)
    lst = tmp_path / 'd.txt'
    lst.write_text('%s 1-5 11-15\n' % (tmp_path / 'clip.npy'))
    ds = vdata.DisjointVideoClipDataset(1, str(lst), 5, 5, (24, 32), (0, 0))
    it = ds[0]
    assert it['clip_label'] == 'clip.npy_1-5_11-15'                           # (:241)
    x = it['targets']
    assert x.shape == (10, 1, 24, 32)
    idx = list(range(0, 5)) + list(range(10, 15))
    bgr = torch.from_numpy(fr[idx][:, :, :, ::-1].copy()).permute(0, 3, 1, 2).float() / 255 * 2 - 1
    assert torch.allclose(x, bgr2gray(bgr), atol=1e-6)
    lst.write_text('%s 1-5\n' % (tmp_path / 'clip.npy'))
    with pytest.raises(RuntimeError, match='format'):                         # (:219-222)
        vdata.DisjointVideoClipDataset(1, str(lst), 5, 5, (24, 32), (0, 0))[0]


def test_error_behaviour(tmp_path):
    lst = tmp_path / 'l.txt'
    lst.write_text('%s 1-4\n' % (tmp_path / 'missing'))
    ds = vdata.ContiguousVideoClipDataset(3, str(lst), 4, False, False, (8, 8), False, (0, 0))
    with pytest.warns(UserWarning), pytest.raises(RuntimeError, match='could not be opened'):       # (:163-166)
        ds[0]
    d = _png_dir(str(tmp_path), 'short', _frames(T=3, H=8, W=8))
    lst.write_text('%s 1-3\n' % d)
    with pytest.raises(RuntimeError, match='too short'):                                               # (:176-180)
        vdata.ContiguousVideoClipDataset(3, str(lst), 4, False, False, (8, 8), False, (0, 0))[0]


def test_resize_is_opencv_style_bilinear():
    f = np.zeros((2, 2, 3), np.uint8)
    f[0, 0], f[0, 1], f[1, 0], f[1, 1] = 0, 100, 200, 40
    out = vdata.resize_bilinear(f, 4, 4)
    # half-pixel centres: output x = 0, 1, 2, 3 sample source -0.25, 0.25, 0.75, 1.25 (edge-clamped)
    row0 = [0, 25, 75, 100]
    assert out[0, :, 0].tolist() == row0
    assert out[1, 0, 0] == 50 and out[3, 3, 0] == 40
    assert vdata.resize_bilinear(f, 2, 2) is f
    big = _frames(T=1, H=48, W=64)[0]
    assert vdata.resize_bilinear(big, 24, 32).shape == (24, 32, 3)


def test_flip_and_backwards_augmentation(tmp_path, monkeypatch):
    fr = _frames(T=6, H=8, W=8)
    np.save(tmp_path / 'c.npy', fr)
    lst = tmp_path / 'l.txt'
    lst.write_text('%s 1-6\n' % (tmp_path / 'c.npy'))
    ds = vdata.ContiguousVideoClipDataset(3, str(lst), 6, True, True, (8, 8), False, (0, 0))
    monkeypatch.setattr(ds.rng, 'random', lambda: 0.9)                        # both augmentations on (:66-67)
    x = ds[0]['targets']
    want = torch.from_numpy(fr[::-1, :, ::-1, ::-1].copy()).permute(0, 3, 1, 2).float() / 255 * 2 - 1
    assert torch.allclose(x, want, atol=1e-6)


def test_random_draws_are_private_and_reproducible(tmp_path):
    """Window starts, augmentation flips and replacement lines come from the dataset's own generator: the same seed gives
    the same clips, and the global random / numpy.random streams (which data-parallel ranks share with nothing) are not
    touched -- a decode retry on one rank cannot desynchronise anything else."""
    import random
    fr = _frames(T=12, H=8, W=8)
    np.save(tmp_path / 'c.npy', fr)
    lst = tmp_path / 'l.txt'
    lst.write_text('%s 1-12\n%s\n%s 3-11\n' % (tmp_path / 'c.npy', tmp_path / 'missing.npy', tmp_path / 'c.npy'))
    mk = lambda seed: vdata.ContiguousVideoClipDataset(3, str(lst), 5, True, True, (8, 8), True, (0, 0), seed=seed)
    random.seed(123); np.random.seed(123)
    g_py, g_np = random.getstate(), np.random.get_state()[1].copy()
    a, b, c = mk(7), mk(7), mk(8)
    xa = [a[i]['targets'] for i in (0, 1, 2, 1)]            # index 1 cannot be opened: replaced by a random other line
    xb = [b[i]['targets'] for i in (0, 1, 2, 1)]
    xc = [c[i]['targets'] for i in (0, 1, 2, 1)]
    assert all(torch.equal(p, q) for p, q in zip(xa, xb))
    assert not all(torch.equal(p, q) for p, q in zip(xa, xc))
    assert random.getstate() == g_py and np.array_equal(np.random.get_state()[1], g_np)
    rec = vdata.ClipRecord('/x/vid.avi 3-7 10-12')
    assert rec.path == '/x/vid.avi' and rec.spans == [(2, 6), (9, 11)] and rec.label(rec.spans) == 'vid.avi_3-7_10-12'
    assert vdata.ClipRecord('/x/v').spans is None
    with pytest.raises(RuntimeError):
        vdata.ClipRecord('/x/v 3_7')


def test_loader_workers_draw_fresh_reproducible_streams_every_epoch(tmp_path):
    """train.py re-iterates its DataLoader every epoch, so the workers are re-created every epoch: worker_init must give
    them a stream that differs between epochs and workers (the reference reseeds the global ``random`` from the loader's
    per-epoch base seed) and is reproducible from the loader's generator + the dataset's seed."""
    fr = _frames(T=40, H=8, W=8)
    np.save(tmp_path / 'c.npy', fr)
    lst = tmp_path / 'l.txt'
    lst.write_text(''.join('%s 1-40\n' % (tmp_path / 'c.npy') for _ in range(8)))

    def run(loader_seed, ds_seed=3):
        ds = vdata.ContiguousVideoClipDataset(3, str(lst), 5, True, True, (8, 8), False, (0, 0), seed=ds_seed)
        g = torch.Generator().manual_seed(loader_seed)
        loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, num_workers=2, worker_init_fn=ds.worker_init, generator=g)
        return [torch.cat([b['targets'] for b in loader]) for _ in range(2)]            # two epochs
    e0, e1 = run(11)
    assert e0.shape == (8, 5, 3, 8, 8)
    assert not torch.equal(e0, e1)                          # a new epoch replays nothing
    assert not torch.equal(e0[0:2], e0[2:4])                # the two workers draw differently (same line, different windows)
    f0, f1 = run(11)
    assert torch.equal(e0, f0) and torch.equal(e1, f1)      # reproducible
    assert not torch.equal(run(12)[0], e0)                  # the loader's generator matters (train.py: opt.seed + 7 * rank)
    assert not torch.equal(run(11, ds_seed=4)[0], e0)       # and so does the dataset's own seed
