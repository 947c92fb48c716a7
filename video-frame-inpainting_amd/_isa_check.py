"""Invariants of the GENERATED device code that the library's correctness depends on and that a different compiler could break
without any source change: checked on the assembly of the very compile that produces the library (_native.build keeps
it with -save-temps=obj) and refused if violated; tests/test_wino_isa.py runs the same checks and feeds them doctored text.

wino43::conv3x3_gen and conv3x3_wrw_gen (the F(4x4, 3x3) kernel and its weight-gradient form): the chunk loop is one asm statement with fixed registers v0-v99 and
   a[0:143]; the compiler must give the kernel no scratch, no spill, and registers for two waves per SIMD (<= 256 in all): a
   spill would be vector-memory traffic the statement's own s_waitcnt vmcnt counting does not expect.
   (Round 4's compiler-scheduled forms of that kernel, whose asm loads depended on the compiler not touching their destination registers
   before a wait it could not see, are gone: every load of the kernel is inside the statement now.)
"""
import re

_KERNEL = re.compile(r'^(_Z\w+):', re.M)


def kernels(asm, prefix):
    """{mangled name: (body lines, descriptor text)} of the kernels whose mangled name starts with ``prefix``."""
    out = {}
    for m in _KERNEL.finditer(asm):
        name = m.group(1)
        if not name.startswith(prefix) or name in out:
            continue
        end = asm.find('.end_amdhsa_kernel', m.start())
        if end < 0:
            continue
        body = asm[m.start():end]
        cut = body.find('.amdhsa_kernel')
        out[name] = (body[:cut].splitlines(), body[cut:])
    return out


def _regs(text):
    """vector registers an operand string mentions"""
    found = set(int(n) for n in re.findall(r'\bv(\d+)\b', text))
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', text):
        found.update(range(int(a), int(b) + 1))
    return found


def check(asm):
    """-> list of violations (empty: the library may be installed)."""
    bad = []
    gen = kernels(asm, '_ZN6wino4311conv3x3_gen')
    if not gen:
        bad.append('no wino43::conv3x3_gen kernel in the assembly')
    wrw = kernels(asm, '_ZN6wino4315conv3x3_wrw_gen')       # the weight-gradient form of the same statement: the same rules
    if not wrw:
        bad.append('no wino43::conv3x3_wrw_gen kernel in the assembly')
    gen.update(wrw)
    for name, (lines, desc) in gen.items():
        get = lambda key: int(re.search(r'\.amdhsa_%s\s+(\d+)' % key, desc).group(1))
        if get('private_segment_fixed_size') != 0 or any(re.search(r'\bscratch_(load|store)', l) for l in lines):
            bad.append('%s: scratch memory in use (spills next to the asm chunk loop)' % name)
        if get('next_free_vgpr') > 256:
            bad.append('%s: %d registers: two waves per SIMD do not fit' % (name, get('next_free_vgpr')))
        if get('next_free_vgpr') - get('accum_offset') < 144:
            bad.append('%s: fewer than 144 accumulator registers' % name)
        n = sum(1 for l in lines if 'v_mfma_f32_16x16x4_f32' in l)
        var = re.match(r'_ZN6wino4311conv3x3_genILin?\d+ELin?\d+ELi(\d+)E', name)
        if var and var.group(1) != '0':      # <ACT, EPI, VAR != 0>: a timing ablation of the tools build (parts of the loop left out on purpose)
            continue
        # chunk bodies of 36 MFMAs: per role one with the next chunk's loads / DMA and the plain one; the weight-gradient form's third role
        # transforms between the MFMA groups and has a third body (a next chunk, none behind it)
        want = 252 if name in wrw else 216
        if n != want:
            bad.append('%s: %d MFMAs in the chunk loops, expected %d' % (name, n, want))
    return bad
