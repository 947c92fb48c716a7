"""Prints one kernel's ISA from build/*.s: python tools/isa_fn.py <mangled-name-substring> [first-line last-line]"""
import sys, collections
s = open('build/sepconv_capi-hip-amdgcn-amd-amdhsa-gfx950.s').read()
name = sys.argv[1]
i = s.find(name + ':') if (name + ':') in s else s.find(name)
i = s.rfind('\n', 0, i) + 1
j = s.find('.end_amdhsa_kernel', i)
L = s[i:j].splitlines()
if len(sys.argv) >= 4:
    print('\n'.join('%5d %s' % (n, l) for n, l in enumerate(L) if int(sys.argv[2]) <= n < int(sys.argv[3])))
else:
    mf = [n for n, l in enumerate(L) if 'v_mfma' in l]
    print('lines', len(L), 'mfma first/last', mf[0] if mf else None, mf[-1] if mf else None)
    for n, l in enumerate(L):
        if l.startswith('.LBB') or 's_cbranch' in l or 's_barrier' in l or 'scratch_' in l: print(n, l)
