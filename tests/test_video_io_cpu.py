"""video_io.py: the AVI demultiplexer and the PIL sequence reader behind data.open_frame_source (the reader interface the
reference takes from imageio: get_length / get_data, src/data/base_dataset.py:125-137), on files synthesised here."""
import io
import os
import struct

import numpy as np
import pytest
import torch
from PIL import Image

from video_frame_inpainting_amd import data as vdata
from video_frame_inpainting_amd import video_io


def _chunk(fourcc, payload):
    return fourcc + struct.pack('<I', len(payload)) + payload + (b'\x00' if len(payload) & 1 else b'')


def _lst(kind, payload):
    return _chunk(b'LIST', kind + payload)


def write_avi(path, frames, codec, bits=24, top_down=False, audio_stream_first=False, drop=()):
    """A minimal AVI 1.0 file: [hdrl [avih] [strl strh strf]...] [movi 00dc...]; codec b'MJPG', b'MPNG' or raw (b'\\0\\0\\0\\0')."""
    T, H, W, _ = frames.shape
    payloads = []
    for t in range(T):
        if t in drop:
            payloads.append(b'')
            continue
        fr = frames[t]
        if codec in (b'MJPG', b'MPNG'):
            buf = io.BytesIO()
            Image.fromarray(fr).save(buf, format='JPEG' if codec == b'MJPG' else 'PNG', quality=95)
            payloads.append(buf.getvalue())
        else:
            if bits == 8:
                pix = fr[:, :, 0:1]                               # gray frames, identity palette
            elif bits == 24:
                pix = fr[:, :, ::-1]
            else:
                pix = np.concatenate([fr[:, :, ::-1], np.full((H, W, 1), 255, np.uint8)], axis=2)
            row = (W * bits // 8 + 3) & ~3
            rows = np.zeros((H, row), np.uint8)
            rows[:, :W * bits // 8] = pix.reshape(H, -1)
            payloads.append((rows if top_down else rows[::-1]).tobytes())
    vid = 1 if audio_stream_first else 0
    comp = codec if codec in (b'MJPG', b'MPNG') else b'\x00\x00\x00\x00'
    bih = struct.pack('<IiiHH4sIiiII', 40, W, -H if top_down else H, 1, bits, comp, 0, 0, 0, 256 if bits == 8 else 0, 0)
    if bits == 8:
        bih += b''.join(struct.pack('<BBBB', i, i, i, 0) for i in range(256))
    strh = lambda t: _chunk(b'strh', t + comp + b'\x00' * 48)
    strls = []
    if audio_stream_first:
        strls.append(_lst(b'strl', strh(b'auds') + _chunk(b'strf', b'\x01\x00' + b'\x00' * 14)))
    strls.append(_lst(b'strl', strh(b'vids') + _chunk(b'strf', bih)))
    hdrl = _lst(b'hdrl', _chunk(b'avih', b'\x00' * 56) + b''.join(strls))
    movi = b''
    for t, p in enumerate(payloads):
        if audio_stream_first:
            movi += _chunk(b'00wb', b'\x01\x02\x03')              # odd length: exercises the padding byte
        movi += _chunk(('%02ddc' % vid).encode() if comp != b'\x00\x00\x00\x00' else ('%02ddb' % vid).encode(), p)
    body = b'AVI ' + hdrl + _chunk(b'JUNK', b'\x00' * 7) + _lst(b'movi', movi) + _chunk(b'idx1', b'')
    with open(path, 'wb') as f:
        f.write(b'RIFF' + struct.pack('<I', len(body)) + body)


def _frames(T=6, H=18, W=22, seed=0):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = np.empty((T, H, W, 3), np.uint8)
    for t in range(T):
        out[t, :, :, 0] = (yy * 9 + t * 20) % 256
        out[t, :, :, 1] = (xx * 7 + t * 5) % 256
        out[t, :, :, 2] = rng.randint(0, 256)
    return out


@pytest.mark.parametrize('bits, top_down', [(24, False), (24, True), (32, False), (8, False)])
def test_uncompressed_avi_frames_are_exact(tmp_path, bits, top_down):
    fr = _frames(W=22 if bits != 8 else 21)                       # width 21 at 8 bits: row padding
    if bits == 8:
        fr[:, :, :, 1] = fr[:, :, :, 0]; fr[:, :, :, 2] = fr[:, :, :, 0]
    p = str(tmp_path / 'raw.avi')
    write_avi(p, fr, b'\x00\x00\x00\x00', bits=bits, top_down=top_down, audio_stream_first=True)
    v = video_io.open_video_file(p)
    assert v.get_length() == fr.shape[0]
    for t in range(fr.shape[0]):
        got = v.get_data(t)
        assert got.dtype == np.uint8 and np.array_equal(got, fr[t]), t
    with pytest.raises(IndexError):
        v.get_data(fr.shape[0])


def test_png_and_mjpeg_avi_and_dropped_frames(tmp_path):
    fr = _frames()
    p = str(tmp_path / 'png.avi')
    write_avi(p, fr, b'MPNG', drop=(3,))
    v = video_io.open_video_file(p)
    assert v.get_length() == 6
    assert np.array_equal(v.get_data(2), fr[2]) and np.array_equal(v.get_data(4), fr[4])
    assert np.array_equal(v.get_data(3), fr[2])                   # a zero-length chunk repeats the previous picture
    smooth = np.repeat(np.repeat(_frames(H=9, W=11), 4, axis=1), 4, axis=2)     # blocky content: JPEG is near-exact on it
    p = str(tmp_path / 'mjpg.avi')
    write_avi(p, smooth, b'MJPG')
    v = video_io.open_video_file(p)
    for t in range(smooth.shape[0]):
        want = np.asarray(Image.open(io.BytesIO(_jpeg(smooth[t]))).convert('RGB'))
        assert np.array_equal(v.get_data(t), want)                # exactly what PIL decodes from that frame's JPEG
        assert np.abs(v.get_data(t).astype(int) - smooth[t]).mean() < 6


def _jpeg(frame):
    buf = io.BytesIO()
    Image.fromarray(frame).save(buf, format='JPEG', quality=95)
    return buf.getvalue()


def test_inter_frame_codecs_are_refused_with_their_fourcc(tmp_path):
    fr = _frames()
    p = str(tmp_path / 'divx.avi')
    write_avi(p, fr, b'MPNG')
    raw = bytearray(open(p, 'rb').read())
    i = raw.find(b'MPNG', raw.find(b'strf'))
    raw[i:i + 4] = b'DX50'                                        # the KTH distribution's codec
    open(p, 'wb').write(bytes(raw))
    with pytest.raises(IOError, match='DX50'):
        video_io.open_video_file(p)
    with pytest.warns(UserWarning, match='DX50'):
        assert vdata.open_frame_source(p) is None                 # the dataset's contract: None, with a warning (:119-122)
    q = str(tmp_path / 'not_a_video.bin')
    open(q, 'wb').write(b'hello world, this is not a container')
    with pytest.warns(UserWarning):
        assert vdata.open_frame_source(q) is None


def test_gif_sequence_and_dataset_pipeline_on_an_avi(tmp_path):
    fr = _frames(T=8, H=24, W=32)
    gif = str(tmp_path / 'clip.gif')
    imgs = [Image.fromarray(np.repeat(f[:, :, :1], 3, axis=2)) for f in fr]          # gray: survives the GIF palette exactly
    imgs[0].save(gif, save_all=True, append_images=imgs[1:], duration=40, loop=0)
    v = video_io.open_video_file(gif)
    assert v.get_length() == 8
    assert np.array_equal(v.get_data(5)[:, :, 0], fr[5][:, :, 0])
    # the reference's dataset over a list that names an AVI file (base_dataset.py:147-202)
    avi = str(tmp_path / 'person01_boxing_d1.avi')
    write_avi(avi, fr, b'\x00\x00\x00\x00')
    lst = tmp_path / 'list.txt'
    lst.write_text('%s 2-8\n' % avi)
    ds = vdata.ContiguousVideoClipDataset(3, str(lst), 7, False, False, (24, 32), False, (0, 0))
    it = ds[0]
    assert it['clip_label'] == 'person01_boxing_d1.avi_2-8'
    want = torch.from_numpy(fr[1:8, :, :, ::-1].copy()).permute(0, 3, 1, 2).float() / 255 * 2 - 1
    assert torch.allclose(it['targets'], want, atol=1e-6)


@pytest.mark.parametrize('bits, cut_to', [(24, 8), (8, 30), (8, 40 + 4 * 100)])
def test_a_truncated_stream_header_warns_and_returns_none(tmp_path, bits, cut_to):
    """a corrupt AVI must not kill a DataLoader worker with struct.error.  The stream format chunk is cut to 8 bytes (no
    BITMAPINFOHEADER), to 30 bytes (8-bit: no colour count) and inside the palette: open_frame_source warns and returns None, as
    the reference does for an unreadable video (src/data/base_dataset.py:118-127) -- the dataset then samples another clip."""
    good = tmp_path / 'good.avi'
    write_avi(str(good), _frames(), b'\x00\x00\x00\x00', bits=bits)
    blob = good.read_bytes()
    at = blob.index(b'strf')
    size = struct.unpack('<I', blob[at + 4:at + 8])[0]
    assert cut_to < size
    # keep the chunk's declared size consistent with what is left of it (a cleanly truncated chunk), and also try the declared
    # size left as it was (the parser then sees the following chunks' bytes as header: still no exception type other than IOError)
    for keep_declared in (False, True):
        head = blob[:at + 4] + (blob[at + 4:at + 8] if keep_declared else struct.pack('<I', cut_to)) + blob[at + 8:at + 8 + cut_to]
        bad = tmp_path / ('bad_%d_%d.avi' % (cut_to, keep_declared))
        bad.write_bytes(head if keep_declared else head + blob[at + 8 + size:])
        with pytest.raises(IOError):
            video_io.open_video_file(str(bad))
        with pytest.warns(UserWarning, match='Failed to open video'):
            assert vdata.open_frame_source(str(bad)) is None
    assert vdata.open_frame_source(str(good)).get_length() == 6
