"""Invariants of the GENERATED device code that the library's correctness depends on and that a different compiler could break
without any source change: checked on the assembly of the very compile that produces the library (_native.build keeps
it with -save-temps=obj) and refused if violated; tests/test_wino_isa.py runs the same checks and feeds them doctored text.

1. wino43::conv3x3_gen (the default F(4x4, 3x3) kernel): its chunk loop is one asm statement with fixed registers v0-v99 and
   a[0:143]; the compiler must give the kernel no scratch, no spill, and registers for two waves per SIMD (<= 256 in all): a
   spill would be vector-memory traffic the statement's own s_waitcnt vmcnt counting does not expect.
2. wino43::conv3x3 (round 4's compiler-scheduled forms, kept for A/B): patch edge values are loaded by asm statements whose
   destination the compiler takes for valid at once; the data lands at the s_waitcnt vmcnt(0) behind the chunk's last MFMA.
   Between the load and that wait NO instruction may read the destination register (a spill or a copy would move stale data).
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
        if n != 216:                       # chunk bodies of 36 MFMAs: per role one with the next chunk's loads / DMA and the plain one
            bad.append('%s: %d MFMAs in the chunk loops, expected 216' % (name, n))
    for name, (lines, desc) in kernels(asm, '_ZN6wino437conv3x3I').items():
        pending, in_asm = {}, False
        for n, l in enumerate(lines):
            t = l.strip()
            if 'ASMSTART' in t:
                in_asm = True
                continue
            if 'ASMEND' in t:
                in_asm = False
                continue
            if not t or t.startswith((';', '.')) or t.endswith(':'):
                continue
            if re.match(r's_waitcnt\s+vmcnt\(0\)', t):
                pending.clear()
                continue
            m = re.match(r'buffer_load_dword\s+v(\d+)\s*,(.*)', t)
            if in_asm and m:
                for r in _regs(m.group(2)) & set(pending):
                    bad.append('%s: line %d reads v%d before the wait that lands it: %s' % (name, n, r, t))
                pending[int(m.group(1))] = n
                continue
            ops = t.split(None, 1)
            if len(ops) < 2 or not pending:
                continue
            first, _, rest = ops[1].partition(',')
            reads = _regs(rest) | (_regs(first) if ops[0].startswith(('buffer_store', 'global_store', 'ds_write', 'scratch_store')) else set())
            for r in reads & set(pending):
                bad.append('%s: line %d reads v%d before the wait that lands it: %s' % (name, n, r, t))
            for r in _regs(first) & set(pending):            # overwritten before it landed: the load is dead, no longer pending
                if not ops[0].startswith(('buffer_store', 'global_store', 'ds_write', 'scratch_store')):
                    bad.append('%s: line %d overwrites v%d while its load is in flight: %s' % (name, n, r, t))
    return bad
