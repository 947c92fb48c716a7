"""Ingestion of the PUBLISHED checkpoints' format (bashes/download/download_model_checkpoints.bash:3): torch-0.3.1 legacy
(non-zip) serialisation, pickle protocol 2, written by Python 2.7 -- every dict key and every raw byte payload is a Python-2
``str`` (BINSTRING opcodes), and the validation sums are numpy float64 scalars whose 8 raw bytes travel as such a ``str``.
Python 3 decodes those as ASCII and fails on the first byte >= 0x80; environments.load (reference environments.py:100-115)
retries with latin1.  No published checkpoint is reachable (no network), so the file is synthesised here with a pickler
that writes strings the Python-2 way."""
import io
import os
import pickle
import struct

import numpy as np
import pytest
import torch

import video_frame_inpainting_amd as vfi
from video_frame_inpainting_amd.environments import BaseVideoFillInEnvironment


class _Py2Pickler(pickle._Pickler):
    """The pure-Python pickler with str / bytes written as Python 2 wrote its ``str``: SHORT_BINSTRING / BINSTRING."""

    def _save_py2_str(self, raw):
        if len(raw) < 256:
            self.write(pickle.SHORT_BINSTRING + bytes([len(raw)]) + raw)
        else:
            self.write(pickle.BINSTRING + struct.pack('<i', len(raw)) + raw)

    def save_str(self, obj):
        self._save_py2_str(obj.encode('latin1'))
        self.memoize(obj)

    def save_bytes(self, obj):
        self._save_py2_str(obj)
        self.memoize(obj)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[str] = save_str
    dispatch[bytes] = save_bytes


class _Py2PickleModule(object):
    Pickler = _Py2Pickler

    @staticmethod
    def dump(obj, f, protocol=2):
        _Py2Pickler(f, protocol).dump(obj)


def _write_py2_style(obj, path):
    torch.save(obj, path, pickle_module=_Py2PickleModule, pickle_protocol=2, _use_new_zipfile_serialization=False)


def test_python2_checkpoint_loads_through_the_latin1_retry(tmp_path):
    torch.manual_seed(4)
    model = vfi.TAIFillInModel(4, 1, 3, 51, num_block=5, kf_dim=2)
    psnr = np.float64(-1234.5678e-3)                       # its IEEE bytes contain values >= 0x80
    assert any(b >= 0x80 for b in psnr.tobytes())
    snapshot = {'updates': 200000, 'sum_avg_psnr_err': psnr, 'sum_avg_ssim_err': np.float64(0.9594),
                'generator': model.state_dict()}
    os.makedirs(tmp_path / 'exp')
    path = tmp_path / 'exp' / 'model_best.ckpt'
    _write_py2_style(snapshot, str(path))
    raw = open(path, 'rb').read()
    assert b'Ugenerator' in raw or b'U\tgenerator' in raw          # the key is a BINSTRING, not a BINUNICODE
    with pytest.raises(UnicodeDecodeError):                       # what a plain Python-3 load makes of it
        torch.load(str(path), map_location='cpu', weights_only=False)

    fresh = vfi.TAIFillInModel(4, 1, 3, 51, num_block=5, kf_dim=2)
    env = BaseVideoFillInEnvironment(fresh, str(tmp_path), 'exp', [0, 0], device='cpu')
    snap = env.load('model_best.ckpt')
    assert snap['updates'] == 200000 and float(snap['sum_avg_psnr_err']) == float(psnr)
    want = model.state_dict()
    got = env.generator.state_dict()
    assert set(got) == set(want)
    assert all(torch.equal(got[k], want[k]) for k in want)
