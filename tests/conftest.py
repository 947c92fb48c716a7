import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    import video_frame_inpainting_amd as vfi
    vfi.configure_miopen()          # the package no longer sets MIOPEN_FIND_MODE on import; the test process asks for FAST itself
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _native_library_built():
    """The C-ABI library is a build artefact (git-ignored): compile it for gfx950 if it is missing or stale.  hipcc
    cross-compiles without a GPU, so this also holds for the CPU-only test run."""
    import shutil
    from video_frame_inpainting_amd import _native
    if shutil.which('hipcc'):
        _native.build()
    yield


class DispatchAt(object):
    """Make conv_ops choose kernels for a batch of ``scale`` x the clips actually given: the workgroup-count thresholds of
    ``_wino_ok`` / ``_wino43_ok`` see N * scale.  With B = 2 and scale 16 every layer takes the route it takes at the 32 clips per
    GPU of configs[1] / configs[2] (Winograd F(2x2) / F(4x4) / the weight-gradient kernel instead of MIOpen on the wide
    low-resolution layers), at a size the CPU oracle finishes in seconds.  ``routes`` counts the decisions."""

    def __init__(self, monkeypatch, scale):
        from video_frame_inpainting_amd import conv_ops
        self.routes = {'wino': 0, 'wino43': 0, 'refused': 0}
        ok, ok43 = conv_ops._wino_ok, conv_ops._wino43_ok

        def wino_ok(N, *a, **k):
            r = ok(N * scale, *a, **k)
            self.routes['wino' if r else 'refused'] += 1
            return r

        def wino43_ok(N, *a, **k):
            r = ok43(N * scale, *a, **k)
            self.routes['wino43'] += bool(r)
            return r
        monkeypatch.setattr(conv_ops, '_wino_ok', wino_ok)
        monkeypatch.setattr(conv_ops, '_wino43_ok', wino43_ok)


@pytest.fixture
def dispatch_at_32_clips(monkeypatch):
    """B = 2 tests through the kernel routes of a 32-clip batch."""
    return DispatchAt(monkeypatch, 16)


def miopen_convolutions(prof):
    """(op name, input shapes) of every ATen convolution (forward or backward: MIOpen on ROCm) a torch.profiler run recorded."""
    names = ('aten::convolution', 'aten::_convolution', 'aten::miopen_convolution', 'aten::convolution_backward',
             'aten::miopen_convolution_backward', 'aten::conv2d', 'aten::miopen_convolution_transpose', 'aten::conv_transpose2d')
    return [(e.name, [tuple(s) for s in e.input_shapes if s]) for e in prof.events() if e.name in names]
