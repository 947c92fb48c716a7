"""Process-level settings are made by entry points, not by importing the package, and every rank of a multi-rank run
gets its own MIOpen find-db / kernel-cache directory."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    e = {k: v for k, v in os.environ.items() if not k.startswith('MIOPEN_') and k not in ('WORLD_SIZE', 'LOCAL_RANK', 'RANK')}
    e.update(env)
    return subprocess.check_output([sys.executable, '-c', 'import sys; sys.path.insert(0, %r)\n' % ROOT + code], env=e, text=True)


def test_import_leaves_the_environment_alone():
    out = _run("import os, video_frame_inpainting_amd\nprint(sorted(k for k in os.environ if k.startswith('MIOPEN_')))")
    assert out.strip() == '[]'


def test_configure_miopen_defaults_and_per_rank_directories(tmp_path):
    code = ("import os, video_frame_inpainting_amd as vfi\nmade = vfi.configure_miopen()\n"
            "print(os.environ.get('MIOPEN_FIND_MODE'), os.environ.get('MIOPEN_USER_DB_PATH'), os.environ.get('MIOPEN_CUSTOM_CACHE_DIR'))")
    single = _run(code, XDG_CACHE_HOME=str(tmp_path)).split()
    assert single == ['FAST', 'None', 'None']
    seen = set()
    for r in range(2):
        mode, db, cache = _run(code, XDG_CACHE_HOME=str(tmp_path), WORLD_SIZE='2', LOCAL_RANK=str(r)).split()
        assert mode == 'FAST' and os.path.isdir(db) and os.path.isdir(cache) and ('rank%d' % r) in db and ('rank%d' % r) in cache
        seen |= {db, cache}
    assert len(seen) == 4                        # nothing shared between the ranks
    # explicit settings win
    mode, db, cache = _run(code, WORLD_SIZE='2', LOCAL_RANK='1', MIOPEN_FIND_MODE='NORMAL', MIOPEN_USER_DB_PATH='/x', XDG_CACHE_HOME=str(tmp_path)).split()
    assert (mode, db) == ('NORMAL', '/x') and 'rank1' in cache
