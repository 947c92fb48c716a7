"""MI355X-native bi-TAI hot path (drop-in for the bi-TAI path of MichiganCOG/video-frame-inpainting).

Layout (mirrors the reference's modules for this path):
  csrc/ + libtai_sepconv.so   HIP kernels behind the C ABI of include/tai_sepconv.h
  separable_convolution       the autograd op (reference src/separable_convolution/SeparableConvolution.py)
  mcnet, tai, create_model    the model (reference src/models/{mcnet,tai}/, create_model.py)
  environments, options       eval / training environments and flags (reference src/environments, src/options)
  losses, sn_discriminator    training-path losses (reference src/losses, src/discriminators)
  metrics, synthetic          PSNR/SSIM definition of the reference; seeded synthetic clips
  graph, parallel             hipGraph capture; clip-sharded data parallelism over RCCL
"""
import os as _os


def configure_miopen(find_mode='FAST', local_rank=None, world=None):
    """Process-level MIOpen settings for an ENTRY POINT (bench.py, train.py, predict.py, the tests' conftest) -- importing the
    package no longer changes the process: call this before the first convolution.

    ``find_mode``: MIOpen (behind torch's convolutions: on this path only shapes the in-tree kernels decline) benchmarks EVERY
    applicable solver the first time it meets a configuration -- reference kernels at 230-260 ms each among them: 22-23 s of the first
    training update, per process, at every start, and a populated user find-db does not shorten it
    (profiles/r04_first_update_find_modes.txt, r04_first_update_find_db.txt).  FAST takes the solver the find-db or MIOpen's heuristic
    names without the search: first update 1.4 s, later updates 2-3 % slower.  An explicit MIOPEN_FIND_MODE in the environment wins.

    ``local_rank`` / ``world`` (default: LOCAL_RANK / WORLD_SIZE of the launcher): with more than one rank per node every rank gets
    its own user find-db and kernel-cache directory, so N first updates never write the same sqlite files concurrently
    (MIOPEN_USER_DB_PATH, MIOPEN_CUSTOM_CACHE_DIR; explicit settings win).  Returns the settings made."""
    made = {}
    if find_mode and 'MIOPEN_FIND_MODE' not in _os.environ:
        made['MIOPEN_FIND_MODE'] = _os.environ['MIOPEN_FIND_MODE'] = find_mode
    world = int(_os.environ.get('WORLD_SIZE', '1')) if world is None else world
    local_rank = int(_os.environ.get('LOCAL_RANK', '0')) if local_rank is None else local_rank
    if world > 1:
        base = _os.path.join(_os.environ.get('XDG_CACHE_HOME') or _os.path.join(_os.path.expanduser('~'), '.cache'), 'tai_miopen')
        for var, sub in (('MIOPEN_USER_DB_PATH', 'db'), ('MIOPEN_CUSTOM_CACHE_DIR', 'cache')):
            if var not in _os.environ:
                path = _os.path.join(base, 'rank%d' % local_rank, sub)
                try:
                    _os.makedirs(path, exist_ok=True)
                except OSError:
                    import tempfile as _tempfile
                    path = _tempfile.mkdtemp(prefix='tai_miopen_rank%d_%s_' % (local_rank, sub))
                made[var] = _os.environ[var] = path
    return made


from .create_model import create_model, supported_model_keys  # noqa: F401,E402
from .separable_convolution import SeparableConvolution  # noqa: F401,E402
from .tai import TAIFillInModel  # noqa: F401,E402
from .mcnet import MCNetFillInModel  # noqa: F401,E402

__version__ = '0.1.0'
