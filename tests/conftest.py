import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
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
