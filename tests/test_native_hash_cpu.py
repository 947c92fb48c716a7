"""The loader refuses a libtai_sepconv.so that was not compiled from the sources next to it (round 3's GPU
memory fault came from a stale binary that survived a failed rebuild).  The library carries the SHA-256 of csrc/*.hip, csrc/*.inc
and include/tai_sepconv.h (tai_sepconv_source_hash()); _native.lib() recomputes it from the tree.  Replaces the unconditional
load of the reference's src/separable_convolution/_ext/cunnex/__init__.py:6-15.  CPU only: nothing is launched."""
import os
import shutil

import pytest

from video_frame_inpainting_amd import _native


def test_the_built_library_carries_the_hash_of_the_tree():
    assert os.path.exists(_native.LIB_PATH), 'run __graft_entry__.build() first'
    have = _native.embedded_source_hash(_native.LIB_PATH)
    assert have is not None and len(have) == 64 and have != 'unknown'
    assert have == _native.source_hash()
    _native.verify(_native.LIB_PATH)          # no raise


@pytest.mark.parametrize('victim', ['wino_conv.hip.inc', 'sepconv_fwd_rowloop.inc', 'sepconv_capi.hip', 'HEADER'])
def test_one_changed_byte_in_any_source_is_refused(tmp_path, victim):
    csrc = tmp_path / 'csrc'
    shutil.copytree(_native.CSRC, csrc)
    header = tmp_path / 'tai_sepconv.h'
    shutil.copy(_native.HEADER, header)
    _native.verify(_native.LIB_PATH, str(csrc), str(header))            # an identical copy of the tree is accepted
    path = header if victim == 'HEADER' else csrc / victim
    data = bytearray(path.read_bytes())
    data[len(data) // 2] ^= 0x01
    path.write_bytes(bytes(data))
    with pytest.raises(_native.NativeLibraryError, match='not built from the sources'):
        _native.verify(_native.LIB_PATH, str(csrc), str(header))


def test_an_added_or_removed_source_is_refused_and_scratch_files_are_not_sources(tmp_path):
    csrc = tmp_path / 'csrc'
    shutil.copytree(_native.CSRC, csrc)
    # the generator's transient output and editor backups do not count (they forced needless rebuilds)
    (csrc / 'sepconv_fwd_rowloop.inc.tmp.12345').write_text('half written')
    (csrc / 'wino_conv.hip.inc~').write_text('backup')
    (csrc / '.wino_conv.hip.inc.swp').write_text('swap')
    _native.verify(_native.LIB_PATH, str(csrc))
    (csrc / 'extra.inc').write_text('// new include\n')
    with pytest.raises(_native.NativeLibraryError):
        _native.verify(_native.LIB_PATH, str(csrc))
    os.remove(csrc / 'extra.inc')
    os.remove(csrc / 'upsample.hip.inc')
    with pytest.raises(_native.NativeLibraryError):
        _native.verify(_native.LIB_PATH, str(csrc))


def test_a_library_without_the_symbol_is_refused(tmp_path):
    """A binary from before this check (or any other .so) has no tai_sepconv_source_hash: refused, not trusted."""
    import subprocess
    src = tmp_path / 'old.c'
    src.write_text('int tai_sepconv_version(void) { return 320; }\n')
    so = tmp_path / 'libold.so'
    subprocess.check_call(['gcc', '-shared', '-fPIC', '-o', str(so), str(src)])
    assert _native.embedded_source_hash(str(so)) is None
    with pytest.raises(_native.NativeLibraryError):
        _native.verify(str(so))


def test_a_failed_compile_leaves_no_library_behind(tmp_path, monkeypatch):
    """build() removes the old binary before it compiles and renames the new one into place only on success."""
    out = tmp_path / 'libtai_sepconv.so'
    out.write_bytes(b'stale')
    monkeypatch.setattr(_native, 'LIB_PATH', str(out))
    bad = tmp_path / 'broken.hip'
    bad.write_text('this is not C++\n')
    monkeypatch.setattr(_native, 'MAIN_SOURCE', str(bad))
    import subprocess
    with pytest.raises(subprocess.CalledProcessError):
        _native.build(force=True)
    assert not out.exists()
    assert not [f for f in os.listdir(tmp_path) if '.building.' in f]
