"""Video containers on the input side of the path, without ffmpeg: an AVI (RIFF) demultiplexer with the intra-frame codecs
that can be decoded with what this image has -- uncompressed DIB frames (8 / 24 / 32 bits per pixel, bottom-up or top-down),
Motion-JPEG (each frame a JPEG: PIL) and PNG-in-AVI -- and animated GIF / multi-page TIFF / APNG / WebP through PIL's
frame seeking.

The reference opens every path with ``imageio.get_reader(path, 'ffmpeg')`` (src/data/base_dataset.py:106-122) and uses two
methods of the reader: ``get_length()`` and ``get_data(i) -> [H, W, 3] uint8 RGB`` (:125-137); the readers here offer
exactly those.  Inter-frame codecs (the KTH distribution's DivX / MPEG-4 ASP, H.264, ...) need a real video decoder: such a
file is refused with the codec's FourCC in the message, and the frame-directory / ``.npy`` sources of data.py remain the
way to feed those clips (``ffmpeg -i clip.avi frames/%04d.png`` on a machine that has it).
"""
import io
import struct

import numpy as np

_INTRA_JPEG = (b'MJPG', b'mjpg', b'JPEG', b'jpeg', b'AVRn', b'LJPG', b'ljpg')
_INTRA_PNG = (b'MPNG', b'mpng', b'PNG ', b'png ')
_RAW = (b'\x00\x00\x00\x00', b'DIB ', b'RGB ', b'RAW ', b'raw ')


class AviVideo(object):
    """AVI 1.0 / OpenDML demultiplexer over a memory map: finds the first video stream's format (``strf``: a
    BITMAPINFOHEADER), walks the ``movi`` list(s) for that stream's chunks (``NNdc`` / ``NNdb``) and decodes a frame on
    demand.  Frame chunks of length 0 (dropped frames) repeat the previous frame, as players do."""

    def __init__(self, path):
        self._filename = path
        self._buf = np.memmap(path, dtype=np.uint8, mode='r')
        data = self._buf
        if data.size < 12 or bytes(data[0:4]) != b'RIFF' or bytes(data[8:12]) not in (b'AVI ', b'AVIX'):
            raise IOError('%s: not a RIFF AVI file' % path)
        self._frames = []              # (offset, length) of each video chunk
        self._fmt = None               # (width, height, bits, compression fourcc, top_down, palette)
        self._stream = None
        pos = 0
        while pos + 12 <= data.size and bytes(data[pos:pos + 4]) == b'RIFF':      # AVI, then AVIX segments (OpenDML)
            size = struct.unpack('<I', bytes(data[pos + 4:pos + 8]))[0]
            self._walk(pos + 12, min(pos + 8 + size, data.size))
            pos += 8 + size + (size & 1)
        if self._fmt is None:
            raise IOError('%s: no video stream' % path)
        if not self._frames:
            raise IOError('%s: no video frames' % path)
        width, height, bits, comp, _, _ = self._fmt
        if comp in _RAW:
            if bits not in (8, 24, 32):
                raise IOError('%s: uncompressed AVI with %d bits per pixel is not supported' % (path, bits))
        elif comp not in _INTRA_JPEG + _INTRA_PNG:
            raise IOError('%s: video codec %r needs a real decoder (only uncompressed, Motion-JPEG and PNG AVIs are read here); '
                          'extract the frames to a directory of images instead' % (path, comp.decode('latin1')))

    def _walk(self, pos, end):
        data = self._buf
        n_streams = getattr(self, '_n_streams', 0)
        while pos + 8 <= end:
            fourcc = bytes(data[pos:pos + 4])
            size = struct.unpack('<I', bytes(data[pos + 4:pos + 8]))[0]
            body = pos + 8
            if fourcc == b'LIST':
                kind = bytes(data[body:body + 4])
                if kind == b'strl':
                    self._cur_stream = n_streams
                    n_streams += 1
                    self._n_streams = n_streams
                    self._cur_is_video = False
                if kind in (b'hdrl', b'strl', b'movi', b'rec '):
                    self._walk(body + 4, min(body + size, end))
            elif fourcc == b'strh':
                self._cur_is_video = bytes(data[body:body + 4]) == b'vids' and self._stream is None
            elif fourcc == b'strf' and getattr(self, '_cur_is_video', False) and self._fmt is None:
                # a truncated or corrupt header is an I/O error of THIS file (the caller warns and samples another clip, as the
                # reference does, base_dataset.py:118-127), not a struct.error that kills the loader worker
                avail = min(body + size, end, len(data)) - body
                if avail < 20:
                    raise IOError('%s: truncated AVI header (stream format chunk of %d bytes)' % (self._filename, max(avail, 0)))
                (hsize, width, height, planes, bits, comp) = struct.unpack('<IiiHH4s', bytes(data[body:body + 20]))
                palette = None
                if bits == 8:
                    if avail < 36:
                        raise IOError('%s: truncated AVI header (no colour count in an 8-bit stream format)' % self._filename)
                    ncol = struct.unpack('<I', bytes(data[body + 32:body + 36]))[0] or 256
                    if hsize < 40 or hsize + 4 * ncol > avail:
                        raise IOError('%s: truncated AVI header (palette of %d colours does not fit its chunk)' % (self._filename, ncol))
                    pal = np.frombuffer(bytes(data[body + hsize:body + hsize + 4 * ncol]), np.uint8).reshape(-1, 4)
                    palette = pal[:, [2, 1, 0]].copy() if pal.shape[0] else None          # BGRA -> RGB
                self._fmt = (width, abs(height), bits, comp, height < 0, palette)
                self._stream = self._cur_stream
            elif len(fourcc) == 4 and fourcc[2:4] in (b'dc', b'db') and fourcc[0:2].isdigit() and self._stream is not None \
                    and int(fourcc[0:2]) == self._stream:
                self._frames.append((body, size))
            pos = body + size + (size & 1)

    def get_length(self):
        return len(self._frames)

    def get_data(self, index):
        if not 0 <= index < len(self._frames):
            raise IndexError('frame %d of %d in %s' % (index, len(self._frames), self._filename))
        while index > 0 and self._frames[index][1] == 0:      # dropped frame: the previous picture stays on screen
            index -= 1
        off, size = self._frames[index]
        width, height, bits, comp, top_down, palette = self._fmt
        chunk = self._buf[off:off + size]
        if comp in _RAW:
            row = (width * bits // 8 + 3) & ~3                 # DIB rows are padded to 4 bytes
            if size < row * height:
                raise IOError('%s: frame %d is truncated' % (self._filename, index))
            pix = np.asarray(chunk[:row * height]).reshape(height, row)[:, :width * bits // 8]
            if bits == 8:
                frame = palette[pix] if palette is not None else np.repeat(pix[:, :, None], 3, axis=2)
            else:
                frame = pix.reshape(height, width, bits // 8)[:, :, 2::-1]        # BGR(A) -> RGB
            if not top_down:
                frame = frame[::-1]
            return np.ascontiguousarray(frame, dtype=np.uint8)
        from PIL import Image
        with Image.open(io.BytesIO(bytes(chunk))) as img:
            return np.asarray(img.convert('RGB'))


class PilSequenceVideo(object):
    """Animated GIF, multi-page TIFF, APNG, animated WebP: whatever PIL opens with ``n_frames`` > 1."""

    def __init__(self, path):
        from PIL import Image
        self._filename = path
        self._img = Image.open(path)
        self._n = getattr(self._img, 'n_frames', 1)
        if self._n < 2:
            raise IOError('%s: a single image, not a frame sequence' % path)

    def get_length(self):
        return self._n

    def get_data(self, index):
        if not 0 <= index < self._n:
            raise IndexError('frame %d of %d in %s' % (index, self._n, self._filename))
        self._img.seek(index)
        return np.asarray(self._img.convert('RGB'))


def open_video_file(path):
    """A reader for ``path`` by its leading bytes; IOError if the container or codec is not one this module reads."""
    with open(path, 'rb') as f:
        head = f.read(12)
    if head[:4] == b'RIFF' and head[8:12] in (b'AVI ', b'AVIX'):
        return AviVideo(path)
    return PilSequenceVideo(path)
