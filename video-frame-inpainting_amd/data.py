"""Clip I/O on the input side of the path: the reference's video-list datasets (src/data/base_dataset.py).

Same constructors, list-file formats, clip labels and per-frame pipeline -- resize to ``image_size`` (cv2.resize's
default bilinear), RGB -> BGR, optional horizontal flip, zero-pad bottom / right by ``padding_size``, scale to [0, 1]
(torchvision ``to_tensor``), map to [-1, 1] (``fore_transform``), gray if ``c_dim == 1`` (``bgr2gray``); optional
temporal reversal -- and the same ``{'targets': [T, C, H, W], 'clip_label': str}`` items.

What differs is where frames come from.  The reference opens every path with ``imageio.get_reader(path, 'ffmpeg')``
(:106-122); neither imageio nor cv2 exists in this image, so a "video" here is, in this order: a directory of image
files (sorted by name; PNG / JPEG / BMP through PIL), a ``.npy`` file or an ``.npz`` member ``frames`` holding
``[T, H, W, 3]`` uint8 RGB, and only then an imageio reader if imageio can be imported at run time.  The resize is a
numpy restatement of OpenCV's INTER_LINEAR (half-pixel centres, edge clamp, no antialiasing) rounded to nearest; OpenCV
computes it in 11-bit fixed point for uint8, so single pixels may differ from it by one grey level.
"""
import os
import random
import re
from warnings import warn

import numpy as np
import torch
import torch.utils.data as data

from .util import bgr2gray, fore_transform

_IMAGE_EXT = ('.png', '.jpg', '.jpeg', '.bmp')


class _ArrayVideo:
    def __init__(self, frames, name):
        if frames.ndim != 4 or frames.shape[3] != 3 or frames.dtype != np.uint8:
            raise IOError('%s: expected [T, H, W, 3] uint8 frames, found %s %s' % (name, frames.shape, frames.dtype))
        self._frames, self._filename = frames, name

    def get_length(self):
        return self._frames.shape[0]

    def get_data(self, index):
        if not 0 <= index < self._frames.shape[0]:
            raise IndexError('frame %d of %d in %s' % (index, self._frames.shape[0], self._filename))
        return self._frames[index]


class _ImageDirVideo:
    def __init__(self, path):
        self._filename = path
        self._files = sorted(f for f in os.listdir(path) if f.lower().endswith(_IMAGE_EXT))
        if not self._files:
            raise IOError('%s: no image files' % path)

    def get_length(self):
        return len(self._files)

    def get_data(self, index):
        from PIL import Image
        if not 0 <= index < len(self._files):
            raise IndexError('frame %d of %d in %s' % (index, len(self._files), self._filename))
        with Image.open(os.path.join(self._filename, self._files[index])) as img:
            return np.asarray(img.convert('RGB'))


def open_frame_source(path):
    """A reader with ``get_length()`` / ``get_data(i) -> [H, W, 3] uint8 RGB`` (the part of imageio's reader interface
    the reference uses, base_dataset.py:125-137), or None if the path cannot be opened."""
    try:
        if os.path.isdir(path):
            return _ImageDirVideo(path)
        if path.endswith('.npy'):
            return _ArrayVideo(np.load(path, mmap_mode='r'), path)
        if path.endswith('.npz'):
            return _ArrayVideo(np.load(path)['frames'], path)
        import imageio                                              # absent in this image; used when present
        return imageio.get_reader(path, 'ffmpeg')
    except (IOError, OSError, ImportError, KeyError, ValueError) as e:
        warn('Failed to open video %s: %s' % (path, e))
        return None


def resize_bilinear(frame, height, width):
    """cv2.resize(frame, (width, height)) with the default INTER_LINEAR, for [H, W, C] uint8."""
    h, w = frame.shape[:2]
    if (h, w) == (height, width):
        return frame
    def taps(n_out, n_in):
        src = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / float(n_out)) - 0.5
        i0 = np.floor(src).astype(np.int64)
        frac = src - i0
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), frac
    y0, y1, fy = taps(height, h)
    x0, x1, fx = taps(width, w)
    f = frame.astype(np.float64)
    top = f[y0][:, x0] * (1 - fx)[None, :, None] + f[y0][:, x1] * fx[None, :, None]
    bot = f[y1][:, x0] * (1 - fx)[None, :, None] + f[y1][:, x1] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)


class ContiguousVideoClipDataset(data.Dataset):
    """base_dataset.py:17-202: one line per video, ``<path>`` or ``<path> <a>-<b>`` (1-indexed inclusive frame range);
    an item is a random window of ``seq_length`` consecutive frames of that range."""

    def __init__(self, c_dim, video_list_path, seq_length, backwards, flip, image_size, resample_on_fail, padding_size):
        super().__init__()
        self.c_dim = c_dim
        self.backwards = backwards
        self.flip = flip
        self.image_size = image_size
        self.resample_on_fail = resample_on_fail
        self.padding_size = padding_size
        with open(video_list_path, 'r') as f:
            self.files = [line.strip() for line in f.readlines()]
        self.seq_len = seq_length

    def __len__(self):
        return len(self.files)

    def open_video(self, vid_path):
        return open_frame_source(vid_path)

    def get_frame(self, vid, frame_index):
        try:
            return np.asarray(vid.get_data(frame_index))
        except IndexError:
            raise
        except Exception as e:                                      # a decoder's read error (:129-137)
            warn('Failed to read frame %d in %s: %s' % (frame_index, getattr(vid, '_filename', '?'), e))
            return None

    def read_seq(self, vid, frame_indexes, clip_label):
        """base_dataset.py:50-103."""
        targets = []
        flip_flag = self.flip and (random.random() > 0.5)
        back_flag = self.backwards and (random.random() > 0.5)
        for t in frame_indexes:
            frame = self.get_frame(vid, t)
            if frame is None:
                return None
            img = resize_bilinear(frame, self.image_size[0], self.image_size[1])[:, :, ::-1]      # RGB -> BGR (:81)
            if flip_flag:
                img = img[:, ::-1, :]
            # cv2.copyMakeBorder(..., BORDER_CONSTANT, -1) on uint8 saturates the -1 to 0, i.e. -1.0 after the [-1, 1] map
            img = np.pad(img, ((0, self.padding_size[0]), (0, self.padding_size[1]), (0, 0)), mode='constant')
            targets.append(torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255))   # to_tensor
        if back_flag:
            targets = targets[::-1]
        target = fore_transform(torch.stack(targets))               # T x C x H x W in [-1, 1]
        if self.c_dim == 1:
            target = bgr2gray(target)
        return {'targets': target, 'clip_label': clip_label}

    def __getitem__(self, index):
        while True:
            split_line = self.files[index].split()
            if len(split_line) == 1:
                video_file_path, full_range_str = split_line[0], None
            else:
                video_file_path, full_range_str = split_line
            vid = self.open_video(video_file_path)
            if vid is None:
                if not self.resample_on_fail:
                    raise RuntimeError('Video at %s could not be opened' % video_file_path)
                index = np.random.randint(0, len(self.files))
                continue
            if full_range_str is None:
                full_range = (0, vid.get_length() - 1)
            else:
                full_range = tuple(int(d) - 1 for d in full_range_str.split('-'))      # 0-indexed, inclusive
            if full_range[1] - full_range[0] + 1 < self.seq_len:
                if not self.resample_on_fail:
                    raise RuntimeError('Interval %s in video %s is too short' % (str(full_range), video_file_path))
                index = np.random.randint(0, len(self.files))
                continue
            start_index = random.randint(full_range[0], full_range[1] - self.seq_len + 1)
            frame_indexes = range(start_index, start_index + self.seq_len)
            clip_label = '%s_%d-%d' % (os.path.basename(video_file_path), full_range[0] + 1, full_range[1] + 1)
            item = self.read_seq(vid, frame_indexes, clip_label)
            if item is None:
                if not self.resample_on_fail:
                    raise RuntimeError('Failed to sample frames starting at %d in %s' % (start_index, video_file_path))
                index = np.random.randint(0, len(self.files))
                continue
            return item


class DisjointVideoClipDataset(ContiguousVideoClipDataset):
    """base_dataset.py:205-248: ``<path> <a>-<b> <c>-<d>``: the preceding frames a..b and the following frames c..d
    (1-indexed inclusive), nothing in between."""

    def __init__(self, c_dim, video_list_path, K, F, image_size, padding_size):
        super().__init__(c_dim, video_list_path, None, False, False, image_size, False, padding_size)
        self.K = K
        self.F = F

    def __getitem__(self, index):
        m = re.match(r'(.+) (\d+)-(\d+) (\d+)-(\d+)', self.files[index])
        if m is None:
            raise RuntimeError('Expected line from video list to have format "<video_path> <A-B> <C-D>", '
                               'but found line "%s")' % self.files[index])
        video_file_path = m.group(1)
        p_a, p_b, f_a, f_b = (int(v) - 1 for v in m.group(2, 3, 4, 5))
        vid = self.open_video(video_file_path)
        if vid is None:
            raise RuntimeError('Video at %s could not be opened' % video_file_path)
        frame_indexes = list(range(p_a, p_b + 1)) + list(range(f_a, f_b + 1))
        clip_label = '%s_%d-%d_%d-%d' % (os.path.basename(video_file_path), p_a + 1, p_b + 1, f_a + 1, f_b + 1)
        item = self.read_seq(vid, frame_indexes, clip_label)
        if item is None:
            raise RuntimeError('Failed to sample frames %d-%d and %d-%d in %s' % (p_a, p_b, f_a, f_b, video_file_path))
        return item
