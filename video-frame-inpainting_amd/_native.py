"""Loader (and in-tree builder) of the C-ABI HIP library ``libtai_sepconv.so``.

The library is the product's only compute backend for the separable convolution: there is no CPU or
eager-PyTorch fallback.  ``lib()`` raises if the shared object is missing or lacks a symbol that
``include/tai_sepconv.h`` declares.
"""
import ctypes
import os
import re
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, 'libtai_sepconv.so')
HEADER = os.path.join(_ROOT, 'include', 'tai_sepconv.h')
CSRC = os.path.join(_PKG, 'csrc')
MAIN_SOURCE = os.path.join(CSRC, 'sepconv_capi.hip')          # the one translation unit; it #includes every *.inc
GENERATED = os.path.join(CSRC, 'sepconv_fwd_rowloop.inc')     # written by tools/gen_fwd_asm.py
GENERATORS = (('gen_fwd_asm.py', GENERATED),                  # (generator under tools/, the include it writes)
              ('gen_wino43_asm.py', os.path.join(CSRC, 'wino43_chunkloop.inc')))


def sources(csrc=None):
    """Every source file under csrc/ (*.hip, *.inc): the staleness check and the source hash cover whatever the translation
    unit includes, listed or not.  Editor backups and the generator's transient '<name>.tmp.<pid>' are not sources."""
    csrc = csrc or CSRC
    return sorted(os.path.join(csrc, f) for f in os.listdir(csrc)
                  if f.endswith(('.hip', '.inc')) and not f.startswith('.') and '.tmp.' not in f)


def source_hash(csrc=None, header=None):
    """SHA-256 over (file name, length, bytes) of every source under csrc/ and of include/tai_sepconv.h, in sorted order: what
    build() compiles into the library (tai_sepconv_source_hash()) and what lib() recomputes from the tree before it trusts a
    binary.  Contents only -- no mtimes, no paths -- so the hash survives the copy to the GPU box."""
    import hashlib
    h = hashlib.sha256()
    for path in sources(csrc) + [header or HEADER]:
        try:
            with open(path, 'rb') as f:
                data = f.read()
        except FileNotFoundError:        # vanished between listdir and open (a concurrent generator run): not a source
            continue
        h.update(os.path.basename(path).encode() + b'\0' + str(len(data)).encode() + b'\0')
        h.update(data)
    return h.hexdigest()


_lib = None


class NativeLibraryError(RuntimeError):
    pass


def declared_symbols():
    """Function names declared in include/tai_sepconv.h."""
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(tai_\w+)\s*\(', text)))


TIMING_LIB_PATH = os.path.join(_ROOT, 'build', 'libtai_sepconv_timing.so')


def build(force=False, verbose=False, timing=False):
    """hipcc --offload-arch=gfx950 -> video-frame-inpainting_amd/libtai_sepconv.so (cross-compiles without a GPU).
    ``timing=True`` builds build/libtai_sepconv_timing.so instead: the same sources with -DTAI_TIMING_VARIANTS, i.e. with
    the timing experiments (forward variants >= 100, Winograd timeline skip levels) that the shipped library leaves out;
    only tools/ links or loads it."""
    out = TIMING_LIB_PATH if timing else LIB_PATH
    regenerate()
    want = source_hash()
    if not force and os.path.exists(out) and embedded_source_hash(out) == want:
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    # the old binary goes first: a compile that fails must leave NOTHING to load (round 3's one GPU fault was a stale library
    # that survived a failed rebuild); the new one is written beside the target and renamed over it only when hipcc succeeded
    if os.path.exists(out):
        os.remove(out)
    # compiled in a directory of its own with -save-temps=obj: the device assembly of THIS compile is checked before the library is
    # installed (_isa_check: properties of the generated code that a different hipcc could break with no source change)
    import shutil
    work = os.path.join(_ROOT, 'build', 'obj.%s.%d' % ('timing' if timing else 'lib', os.getpid()))
    os.makedirs(work, exist_ok=True)
    tmp = os.path.join(work, os.path.basename(out))
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-w', '-save-temps=obj',
           '-I' + os.path.join(_ROOT, 'include'), '-DTAI_SOURCE_HASH="%s"' % want, '-o', tmp, MAIN_SOURCE] + \
          (['-DTAI_TIMING_VARIANTS'] if timing else [])
    if verbose:
        print(' '.join(cmd))
    try:
        subprocess.check_call(cmd)
        if source_hash() != want:
            raise NativeLibraryError('csrc/ changed while %s was compiling: build again' % os.path.basename(out))
        asm = os.path.join(work, 'sepconv_capi-hip-amdgcn-amd-amdhsa-gfx950.s')
        from . import _isa_check
        violations = _isa_check.check(open(asm).read())
        if violations:
            raise NativeLibraryError('the device code hipcc generated for %s breaks an invariant the kernels rely on; the library is NOT '
                                     'installed:\n  %s' % (os.path.basename(out), '\n  '.join(violations[:20])))
        if not timing:          # kept for tools/isa_fn.py and a look at what the compiler did
            os.replace(asm, os.path.join(_ROOT, 'build', 'sepconv_capi-hip-amdgcn-amd-amdhsa-gfx950.s'))
        os.replace(tmp, out)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def embedded_source_hash(path):
    """The hash a built library carries, read from the FILE (the marker 'TAI_SOURCE_HASH=' in front of it): nothing is mapped, so
    a binary about to be refused never runs a constructor, and dlopen's by-name cache cannot answer for a file that was rebuilt
    since.  None if the file is missing or carries no hash (any library from before this check)."""
    try:
        with open(path, 'rb') as f:
            data = f.read()
    except OSError:
        return None
    m = re.search(rb'TAI_SOURCE_HASH=([0-9a-f]{64})\0', data)
    return m.group(1).decode() if m else None


def verify(path, csrc=None, header=None):
    """Raise NativeLibraryError unless the library at `path` was compiled from exactly the sources in the tree."""
    have, want = embedded_source_hash(path), source_hash(csrc, header)
    if have != want:
        raise NativeLibraryError(
            '%s was not built from the sources next to it (library: %s, tree: %s): a stale binary must not run -- rebuild with '
            '`python -c "import __graft_entry__ as g; g.build()"`' % (path, have, want))


def regenerate():
    """Re-run the asm generators whose script is newer than its output.  A generator writes a temporary file and renames
    it over the target, so a concurrent hipcc never reads a half-written include."""
    for name, out in GENERATORS:
        gen = os.path.join(_ROOT, 'tools', name)
        if os.path.exists(gen) and (not os.path.exists(out) or os.path.getmtime(gen) > os.path.getmtime(out)):
            subprocess.check_call(['python3', gen], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if os.environ.get('TAI_NATIVE_TIMING_LIB') == '1':     # tools/ only: the build with the timing experiments compiled in
        path = TIMING_LIB_PATH
    if not os.path.exists(path):
        raise NativeLibraryError(
            '%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(there is no fallback path for the separable convolution)' % path)
    import torch  # noqa: F401  -- load torch's HIP runtime first so this library binds to the same libamdhip64
    verify(path)
    L = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    if missing:
        raise NativeLibraryError('%s does not export %s' % (path, missing))
    P, I, V = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    L.tai_sepconv_forward.argtypes = [P, P, P, P, I, I, I, I, I, V]
    L.tai_sepconv_forward.restype = I
    L.tai_sepconv_backward.argtypes = [P, P, P, P, P, P, P, I, I, I, I, I, V]
    L.tai_sepconv_backward.restype = I
    L.tai_upsample_bilinear2x_forward.argtypes = [P, P, I, I, I, V]
    L.tai_upsample_bilinear2x_forward.restype = I
    L.tai_upsample_bilinear2x_backward.argtypes = [P, P, I, I, I, V]
    L.tai_upsample_bilinear2x_backward.restype = I
    L.tai_conv3x3_wino_weight_floats.argtypes = [I, I]
    L.tai_conv3x3_wino_weight_floats.restype = ctypes.c_longlong
    L.tai_conv3x3_wino_transform_weights.argtypes = [P, P, I, I, V]
    L.tai_conv3x3_wino_transform_weights.restype = I
    L.tai_conv3x3_wino_forward.argtypes = [P, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv3x3_wino43_weight_floats.argtypes = [I, I]
    L.tai_conv3x3_wino43_weight_floats.restype = ctypes.c_longlong
    L.tai_conv3x3_wino43_transform_weights.argtypes = [P, P, I, I, V]
    L.tai_conv3x3_wino43_transform_weights.restype = I
    L.tai_conv3x3_wino43_forward.argtypes = [P, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv3x3_wino43_forward.restype = I
    L.tai_conv3x3_wino43_forward_parts.argtypes = [P, I, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv3x3_wino43_forward_parts.restype = I
    L.tai_conv3x3_wino43_forward_ex.argtypes = [P, I, P, P, P, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv3x3_wino43_forward_ex.restype = I
    L.tai_conv3x3_wino43_forward_blocks.argtypes = [P, I, P, P, P, P] + [I] * 14 + [V]
    L.tai_conv3x3_wino43_forward_blocks.restype = I
    L.tai_conv3x3_wino43_set_waves.argtypes = [I]
    L.tai_conv3x3_wino43_set_waves.restype = I
    L.tai_conv3x3_wino_forward.restype = I
    L.tai_conv3x3_wino_forward_maxpool.argtypes = [P, P, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv3x3_wino_forward_maxpool.restype = I
    L.tai_conv3x3_wino_forward_window.argtypes = [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, V]
    L.tai_conv3x3_wino_forward_window.restype = I
    L.tai_conv3x3_wino_forward_ex.argtypes = [P, I, I, P, P, P, P, I, I, I, I, P, P] + [I] * 10 + [V]
    L.tai_conv3x3_wino_forward_ex.restype = I
    L.tai_conv_cin1_forward_maxpool_window.argtypes = [P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, V]
    L.tai_conv_cin1_forward_maxpool_window.restype = I
    L.tai_conv3x3_wino_set_tall.argtypes = [I]
    L.tai_conv3x3_wino_set_arithmetic.argtypes = [I]
    L.tai_conv3x3_wino_set_arithmetic.restype = I
    L.tai_conv3x3_wino_get_arithmetic.argtypes = []
    L.tai_conv3x3_wino_forget_weights.argtypes = [P]
    L.tai_conv3x3_wino_forget_weights.restype = I
    L.tai_conv3x3_wino_get_arithmetic.restype = I
    L.tai_conv3x3_wino_set_tall.restype = I
    L.tai_conv3x3_wino_forward_parts.argtypes = [P, I, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv3x3_wino_forward_parts.restype = I
    L.tai_conv3x3_wino_forward_timeline.argtypes = [P, P, P, P, I, I, I, I, I, P, V]
    L.tai_conv3x3_wino_forward_timeline.restype = I
    L.tai_conv_cin1_forward.argtypes = [P, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv_cin1_forward.restype = I
    L.tai_conv_cin1_forward_maxpool.argtypes = [P, P, P, P, P, I, I, I, I, I, I, V]
    L.tai_conv_cin1_forward_maxpool.restype = I
    L.tai_unpool2x_add.argtypes = [P, P, P, ctypes.c_longlong, I, I, V]
    L.tai_unpool2x_add.restype = I
    L.tai_convlstm_gates_forward.argtypes = [P, P, P, P, I, I, I, ctypes.c_float, V]
    L.tai_convlstm_gates_forward.restype = I
    L.tai_convlstm_gates_backward.argtypes = [P, P, P, P, P, P, P, I, I, I, ctypes.c_float, V]
    L.tai_convlstm_gates_backward.restype = I
    L.tai_conv3x3_wino_wrw_workspace_floats.argtypes = [I] * 5
    L.tai_conv3x3_wino_wrw_workspace_floats.restype = ctypes.c_longlong
    L.tai_conv3x3_wino_wrw.argtypes = [P, P, P, P, P, I, I, I, I, I, V]
    L.tai_conv3x3_wino_wrw.restype = I
    L.tai_conv3x3_wino_wrw_window.argtypes = [P, P, P, P, P] + [I] * 9 + [V]
    L.tai_conv3x3_wino_wrw_window.restype = I
    L.tai_conv3x3_wino_wrw_set_paired.argtypes = [I]
    L.tai_conv3x3_wino_wrw_set_paired.restype = I
    L.tai_conv3x3_wino43_set_placement.argtypes = [I]
    L.tai_conv3x3_wino43_set_placement.restype = I
    L.tai_conv3x3_wino_wrw_set_tile.argtypes = [I]
    L.tai_conv3x3_wino_wrw_set_tile.restype = I
    L.tai_window_scale_bias_lrelu.argtypes = [P, P, P, I, I, I, I, ctypes.c_float, V]
    L.tai_window_scale_bias_lrelu.restype = I
    L.tai_window_scale_lrelu_backward.argtypes = [P, P, P, P, P, I, I, I, I, ctypes.c_float, V]
    L.tai_window_scale_lrelu_backward.restype = I
    L.tai_thin_conv_wrw.argtypes = [P, P, P, P, P, I, I, I, I, I, V]
    L.tai_thin_conv_wrw.restype = I
    L.tai_act_maxpool2x2_forward.argtypes = [P, P, P, ctypes.c_longlong, I, I, I, V]
    L.tai_act_maxpool2x2_forward.restype = I
    L.tai_act_maxpool2x2_backward.argtypes = [P, P, P, P, ctypes.c_longlong, I, I, I, V]
    L.tai_act_maxpool2x2_backward.restype = I
    L.tai_sn_power_iteration.argtypes = [P, P, P, P, I, I, I, V]
    L.tai_sn_power_iteration.restype = I
    L.tai_conv_shift_stack.argtypes = [P, P, I, I, I, I, I, V]
    L.tai_conv_shift_stack.restype = I
    L.tai_conv_cout1_3x3_forward.argtypes = [P, P, P, P, I, I, I, I, I, V]
    L.tai_conv_cout1_3x3_forward.restype = I
    L.tai_conv_cout1_5x5_forward.argtypes = [P, P, P, P, I, I, I, I, V]
    L.tai_conv_cout1_5x5_forward.restype = I
    L.tai_bias_act_inplace.argtypes = [P, P, I, I, I, I, V]
    L.tai_bias_act_inplace.restype = I
    L.tai_sepconv_set_forward_variant.argtypes = [I]
    L.tai_sepconv_set_forward_variant.restype = I
    L.tai_sepconv_default_forward_variant.argtypes = [I, I, I]
    L.tai_sepconv_default_forward_variant.restype = I
    L.tai_sepconv_set_grad_taps_variant.argtypes = [I]
    L.tai_sepconv_set_grad_taps_variant.restype = I
    L.tai_sepconv_set_grad_input_variant.argtypes = [I]
    L.tai_sepconv_set_grad_input_variant.restype = I
    L.tai_sepconv_forward_bytes.argtypes = [I] * 5
    L.tai_sepconv_forward_bytes.restype = ctypes.c_longlong
    L.tai_sepconv_backward_bytes.argtypes = [I] * 5
    L.tai_sepconv_backward_bytes.restype = ctypes.c_longlong
    L.tai_hbm_read_probe.argtypes = [P, ctypes.c_longlong, I, P, V]
    L.tai_hbm_read_probe.restype = I
    L.tai_sepconv_last_error.restype = ctypes.c_char_p
    L.tai_sepconv_source_hash.restype = ctypes.c_char_p
    L.tai_sepconv_version.restype = I
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, lib().tai_sepconv_last_error().decode()))
