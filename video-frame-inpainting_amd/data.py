"""Clip I/O on the input side of the path: the reference's video-list datasets (src/data/base_dataset.py).

Same constructors, list-file formats, clip labels and per-frame pipeline -- resize to ``image_size`` (cv2.resize's
default bilinear), RGB -> BGR, optional horizontal flip, zero-pad bottom / right by ``padding_size``, scale to [0, 1]
(torchvision ``to_tensor``), map to [-1, 1] (``fore_transform``), gray if ``c_dim == 1`` (``bgr2gray``); optional
temporal reversal -- and the same ``{'targets': [T, C, H, W], 'clip_label': str}`` items.

What differs is where frames come from.  The reference opens every path with ``imageio.get_reader(path, 'ffmpeg')``
(:106-122); neither imageio nor cv2 exists in this image, so a "video" here is, in this order: a directory of image
files (sorted by name; PNG / JPEG / BMP through PIL), a ``.npy`` file or an ``.npz`` member ``frames`` holding
``[T, H, W, 3]`` uint8 RGB, then an imageio reader if imageio can be imported at run time, else video_io.py's own readers:
AVI files with intra-frame codecs (uncompressed, Motion-JPEG, PNG) and PIL frame sequences (GIF, TIFF, APNG, WebP).
Inter-frame codecs (DivX, H.264) are refused with the FourCC in the message.  The resize is a
numpy restatement of OpenCV's INTER_LINEAR (half-pixel centres, edge clamp, no antialiasing) rounded to nearest; OpenCV
computes it in 11-bit fixed point for uint8, so single pixels may differ from it by one grey level.
"""
import os
import random
import re
from warnings import warn

import struct

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
        try:
            import imageio                                          # absent in this image; used when present
        except ImportError:
            from .video_io import open_video_file                   # AVI (uncompressed / Motion-JPEG / PNG), GIF, TIFF, ...
            return open_video_file(path)
        return imageio.get_reader(path, 'ffmpeg')
    except (IOError, OSError, ImportError, KeyError, ValueError, struct.error) as e:
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


class ClipRecord(object):
    """One parsed line of a video list.  Formats (reference base_dataset.py:147-156, :214-222), frame numbers 1-indexed and
    inclusive in the file, 0-indexed here:
        <path>                      the whole video                      spans = None
        <path> a-b                  one span                             spans = [(a-1, b-1)]
        <path> a-b c-d              preceding and following span         spans = [(a-1, b-1), (c-1, d-1)]"""

    _SPAN = re.compile(r'^(\d+)-(\d+)$')

    def __init__(self, line):
        fields = line.split()
        if not fields:
            raise RuntimeError('Empty line in video list')
        self.line = line
        self.path = fields[0]
        self.spans = None
        if len(fields) > 1:
            matches = [self._SPAN.match(f) for f in fields[1:]]
            if not all(matches):
                raise RuntimeError('Cannot parse frame ranges of video-list line "%s"' % line)
            self.spans = [(int(m.group(1)) - 1, int(m.group(2)) - 1) for m in matches]

    def label(self, spans):
        """``<basename>_a-b[_c-d]`` with 1-indexed inclusive numbers: the directory name predict.py writes into."""
        return '_'.join([os.path.basename(self.path)] + ['%d-%d' % (a + 1, b + 1) for a, b in spans])


class _ClipReader(object):
    """Frames of one clip -> the tensor the models take (base_dataset.py:50-103): resize, RGB -> BGR, optional horizontal
    flip, constant padding on the bottom / right, [-1, 1], gray if c_dim == 1, optional time reversal."""

    def __init__(self, c_dim, image_size, padding_size):
        self.c_dim, self.image_size, self.padding_size = c_dim, image_size, padding_size

    @staticmethod
    def frame(source, index):
        try:
            return np.asarray(source.get_data(index))
        except IndexError:
            raise
        except Exception as e:                                      # a decoder's read error (:129-137)
            warn('Failed to read frame %d in %s: %s' % (index, getattr(source, '_filename', '?'), e))
            return None

    def clip(self, source, indexes, mirror, reverse):
        frames = []
        for t in indexes:
            raw = self.frame(source, t)
            if raw is None:
                return None
            img = resize_bilinear(raw, self.image_size[0], self.image_size[1])[:, :, ::-1]      # RGB -> BGR (:81)
            if mirror:
                img = img[:, ::-1, :]
            # cv2.copyMakeBorder(..., BORDER_CONSTANT, -1) on uint8 saturates the -1 to 0, i.e. -1.0 after the [-1, 1] map
            img = np.pad(img, ((0, self.padding_size[0]), (0, self.padding_size[1]), (0, 0)), mode='constant')
            frames.append(torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255))
        if reverse:
            frames.reverse()
        clip = fore_transform(torch.stack(frames))                  # T x C x H x W in [-1, 1]
        return bgr2gray(clip) if self.c_dim == 1 else clip


class ContiguousVideoClipDataset(data.Dataset):
    """base_dataset.py:17-202: one line per video, ``<path>`` or ``<path> <a>-<b>``; an item is a random window of
    ``seq_length`` consecutive frames of that range, optionally mirrored / reversed; a video that cannot be opened, is too
    short or fails to decode is replaced by another random line when ``resample_on_fail`` (else RuntimeError).

    Every random draw (window start, augmentation coin flips, replacement line) comes from a PRIVATE generator seeded per
    dataset (``seed``; reseeded per DataLoader worker by ``worker_init``): the reference draws from the global ``random``
    and ``numpy.random`` modules, which in a data-parallel run is also where (K, T, F) used to come from -- one rank's
    decode retry would desynchronise every rank's model shapes."""

    def __init__(self, c_dim, video_list_path, seq_length, backwards, flip, image_size, resample_on_fail, padding_size,
                 seed=0):
        super().__init__()
        with open(video_list_path, 'r') as f:
            self.files = [line.strip() for line in f.readlines()]
        self.seq_len = seq_length
        self.backwards, self.flip, self.resample_on_fail = backwards, flip, resample_on_fail
        self.reader = _ClipReader(c_dim, image_size, padding_size)
        self._seed = int(seed)
        self.rng = random.Random(self._seed)

    def worker_init(self, worker_id):
        """DataLoader ``worker_init_fn``.  Inside a worker ``torch.initial_seed()`` is the loader's base seed + worker id,
        and the base seed is drawn afresh from the loader's generator every time the loader is iterated -- so, as in the
        reference (whose workers reseed the global ``random`` from that same base seed), every epoch and every worker
        gets its own stream, reproducible from the loader's generator and this dataset's seed."""
        self.rng = random.Random(int(torch.initial_seed()) ^ (self._seed << 20))

    def __len__(self):
        return len(self.files)

    def open_video(self, vid_path):
        return open_frame_source(vid_path)

    def _try(self, record):
        """-> (item, None) or (None, reason)."""
        source = self.open_video(record.path)
        if source is None:
            return None, 'Video at %s could not be opened' % record.path
        first, last = record.spans[0] if record.spans else (0, source.get_length() - 1)
        if last - first + 1 < self.seq_len:
            return None, 'Interval %s in video %s is too short' % (str((first, last)), record.path)
        start = self.rng.randint(first, last - self.seq_len + 1)
        mirror = self.flip and self.rng.random() > 0.5
        reverse = self.backwards and self.rng.random() > 0.5
        clip = self.reader.clip(source, range(start, start + self.seq_len), mirror, reverse)
        if clip is None:
            return None, 'Failed to sample frames starting at %d in %s' % (start, record.path)
        return {'targets': clip, 'clip_label': record.label([(first, last)])}, None

    def __getitem__(self, index):
        item, reason = self._try(ClipRecord(self.files[index]))
        while item is None:
            if not self.resample_on_fail:
                raise RuntimeError(reason)
            item, reason = self._try(ClipRecord(self.files[self.rng.randrange(len(self.files))]))
        return item


class DisjointVideoClipDataset(ContiguousVideoClipDataset):
    """base_dataset.py:205-248: ``<path> <a>-<b> <c>-<d>``: the preceding frames a..b and the following frames c..d,
    nothing in between; no augmentation, no resampling."""

    def __init__(self, c_dim, video_list_path, K, F, image_size, padding_size):
        super().__init__(c_dim, video_list_path, None, False, False, image_size, False, padding_size)
        self.K, self.F = K, F

    def __getitem__(self, index):
        try:
            record = ClipRecord(self.files[index])
        except RuntimeError:
            record = None
        if record is None or record.spans is None or len(record.spans) != 2:
            raise RuntimeError('Expected line from video list to have format "<video_path> <A-B> <C-D>", '
                               'but found line "%s")' % self.files[index])
        source = self.open_video(record.path)
        if source is None:
            raise RuntimeError('Video at %s could not be opened' % record.path)
        indexes = [t for a, b in record.spans for t in range(a, b + 1)]
        clip = self.reader.clip(source, indexes, False, False)
        if clip is None:
            raise RuntimeError('Failed to sample frames %s in %s' % (record.label(record.spans), record.path))
        return {'targets': clip, 'clip_label': record.label(record.spans)}
