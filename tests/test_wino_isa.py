"""The Winograd kernel's weight DMA (global_load_lds_dwordx4 in inline asm) is invisible to the compiler's wait counting;
its completion is awaited by COUNTED s_waitcnt vmcnt(N) that assume N compiler-tracked loads are issued after the last
DMA of a chunk.  That is a property of the generated code, so it is checked on the generated code: compile the device
side to assembly and count (csrc/wino_conv.hip.inc; profiles/r01_notes.md item 9)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def device_asm(tmp_path_factory):
    if shutil.which('hipcc') is None:
        pytest.skip('hipcc not available')
    kept = os.path.join(ROOT, 'build', 'sepconv_capi-hip-amdgcn-amd-amdhsa-gfx950.s')      # _native.build() keeps the assembly of its compile
    lib = os.path.join(ROOT, 'video-frame-inpainting_amd', 'libtai_sepconv.so')
    if os.path.exists(kept) and os.path.exists(lib) and os.path.getmtime(kept) <= os.path.getmtime(lib) + 1 and \
            abs(os.path.getmtime(kept) - os.path.getmtime(lib)) < 600:
        from video_frame_inpainting_amd import _native
        if _native.embedded_source_hash(lib) == _native.source_hash():
            return open(kept).read()
    out = tmp_path_factory.mktemp('isa') / 'capi.s'
    cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-w', '-I' + os.path.join(ROOT, 'include'),
           '--cuda-device-only', '-S', os.path.join(ROOT, 'video-frame-inpainting_amd', 'csrc', 'sepconv_capi.hip'), '-o', str(out)]
    subprocess.run(cmd, check=True, timeout=900)
    return open(out).read()


def _kernel_bodies(asm, mangled_prefix):
    names = sorted(set(re.findall(r'^(%s\w*):' % re.escape(mangled_prefix), asm, flags=re.M)))
    for name in names:
        i = asm.find(name + ':')
        yield name, asm[i:asm.find('.end_amdhsa_kernel', i)].splitlines()


@pytest.mark.parametrize('tall,dmas', [(0, 8), (1, 16)])
@pytest.mark.parametrize('parts', [0, 1])
def test_counted_waits_match_the_loads_behind_the_last_dma(device_asm, parts, tall, dmas):
    # conv3x3<ACT 1, DBG 0, SKIP 0, PARTS parts, TALL tall> (64 x 64 and 128 x 32 workgroups): the instantiations the product launches
    prefix = '_ZN4wino7conv3x3ILi1ELi0ELi0ELi%dELb%dELi0EE' % (parts, tall)
    found = 0
    for name, lines in _kernel_bodies(device_asm, prefix):
        found += 1
        bars = [n for n, l in enumerate(lines) if 's_barrier' in l]
        segs = [lines[a + 1:b + 1] for a, b in zip([-1] + bars[:-1], bars)]
        dma_segs = [s for s in segs if any('global_load_lds_dwordx4' in l for l in s)]
        assert len(dma_segs) == 5, (name, len(dma_segs))         # prologue, chunk 0 (peeled), the two loop bodies, the trailing chunk
        waits = [[int(m.group(1)) for l in s for m in [re.search(r's_waitcnt vmcnt\((\d+)\)', l)] if m][-1] for s in dma_segs]
        # prologue (round 4): only the weight DMA must have landed at the barrier; the middle pairs of chunk 1's patches are issued
        # behind it and stay in flight -- ONE load per patch row (the outer values go out before the DMA, and only if a lane of the
        # wave needs them), 8 rows (two patches per thread) or 4 (one): the wait must count exactly them
        pro = dma_segs[0]
        last = max(n for n, l in enumerate(pro) if 'global_load_lds_dwordx4' in l)
        behind = sum(1 for l in pro[last:] if re.search(r'\bbuffer_load_dword', l))
        assert behind == waits[0] == (4 if tall else 8), (name, behind, waits[0])
        # every kernel argument of the prologue arrives in ONE batch of scalar loads: a single lgkmcnt wait before the first patch load
        first_load = next(n for n, l in enumerate(pro) if re.search(r'\bbuffer_load_dword', l))
        assert sum(1 for l in pro[:first_load] if re.search(r's_waitcnt.*lgkmcnt', l)) <= 2, name
        # and no load sits between the prologue's barrier and the first MFMA (round 3: 24 of them, 1,300 cycles)
        first_mfma = next(n for n, l in enumerate(dma_segs[1]) if 'v_mfma' in l)
        assert not any(re.search(r'\bbuffer_load', l) for l in dma_segs[1][:first_mfma]), name
        for s, w in zip(dma_segs[1:4], waits[1:4]):
            last = max(n for n, l in enumerate(s) if 'global_load_lds_dwordx4' in l)
            behind = sum(1 for l in s[last:] if re.search(r'\bbuffer_load_dword', l))
            assert sum(1 for l in s if 'global_load_lds_dwordx4' in l) == dmas
            assert behind == w == 12, (name, behind, w)
        # no register spills in the channel loop (they would sit on the MFMA critical path)
        for s in dma_segs[1:4]:
            assert not any(re.search(r'\bscratch_(load|store)', l) for l in s), name
    assert found == 1


@pytest.mark.parametrize('pair', [0, 1])
def test_weight_gradient_chunk_loop_keeps_its_accumulators_in_place(device_asm, pair):
    """The weight-gradient kernel's chunk loop (csrc/wino_wrw.hip.inc) must be one basic block of 128 MFMAs with no spill and
    no accumulator traffic: with branches in the tile-position update the register allocator once copied all 256
    accumulator registers through scratch and VGPRs on every iteration (2x the kernel time) -- a property
    of the generated code, so it is checked on the generated code."""
    prefix = '_ZN4wino3wrw11conv3x3_wrwILi0ELb%dEE' % pair
    found = 0
    for name, lines in _kernel_bodies(device_asm, prefix):
        found += 1
        mfma = [n for n, l in enumerate(lines) if 'v_mfma_f32_32x32x2_f32' in l]
        assert len(mfma) == 128, (name, len(mfma))               # two chunk bodies, nothing peeled or duplicated
        loop = lines[mfma[0]:mfma[-1] + 1]
        assert not any(re.search(r'\bscratch_(load|store)', l) for l in loop), name
        assert not any(re.search(r'\bv_accvgpr_(read|write|mov)', l) for l in loop), name
        labels = [l for l in loop if re.match(r'^\.LBB\w+:', l.strip())]
        assert not labels, (name, labels)                        # no block boundary between the first and the last MFMA
        assert sum(1 for l in loop if 's_barrier' in l) == 1     # (the second chunk's barrier follows its last MFMA)
        # whole-line loads in the paired form: 12 x 16-byte loads per pair of chunks, none in the plain form
        wide = sum(1 for l in loop if 'buffer_load_dwordx4' in l)
        assert wide == (12 if pair else 0), (name, wide)
    assert found == 1


def test_persistent_sepconv_kernel_polls_lds_and_keeps_its_loads_in_flight(device_asm):
    """Properties of the generated code that kernel 20 (csrc/sepconv_fwd.hip.inc, sepconv_forward_persistent) depends on and that
    the compiler once got wrong: (1) the LDS counters are polled with ds_read, not through a generic pointer -- a flat load sits
    behind s_waitcnt vmcnt(0), so every poll would wait for the tap loads and the patch DMA in flight; (2) the type-A tap loads
    are 51 buffer loads off one descriptor with nothing waited for in between (per-lane 64-bit addresses were computed up front,
    spilled and reloaded, one load at a time); (3) next to no register spills."""
    found = 0
    for name, lines in _kernel_bodies(device_asm, '_ZN3fwd26sepconv_forward_persistentILi0E'):
        found += 1
        # <DBG 0, NT, REV>: round 4's instantiations with non-temporal tap loads carry `nt` on every once-read tap stream -- the 51
        # type-A buffer loads, the 51 fold loads and the v planes' LDS-DMA (the 2 x 12 patch DMAs keep the default policy)
        nt = 'Lb1ELb' in name
        for pat, want in ((r'\bbuffer_load_dwordx4\b', 51), (r'\bglobal_load_dwordx4\b', 51)):
            hits = [l for l in lines if re.search(pat, l)]
            assert len(hits) == want and sum(1 for l in hits if re.search(r'\bnt\b', l)) == (want if nt else 0), (name, pat)
        dma = [l for l in lines if 'global_load_lds_dwordx4' in l]
        assert sum(1 for l in dma if re.search(r'\bnt\b', l)) == (len(dma) - 24 if nt else 0), name
        assert not any('flat_load' in l or 'flat_atomic' in l for l in lines), name
        assert sum(1 for l in lines if re.search(r'\bds_add_u32\b', l)) >= 4           # ready / done bumps
        # (a handful of spills around the row-loop blocks, whose register maps leave the compiler seven free vector registers, are
        # tolerated; a spilled tap address array or accumulator set would be hundreds)
        assert sum(1 for l in lines if re.search(r'\bscratch_(load|store)', l)) <= 24, name
        # the 51 tap loads of a round: consecutive buffer_load_dwordx4, no s_waitcnt vmcnt between the first and the last
        idx = [n for n, l in enumerate(lines) if re.search(r'\bbuffer_load_dwordx4\b', l)]
        assert len(idx) == 51, (name, len(idx))
        between = lines[idx[0]:idx[-1] + 1]
        assert not any(re.search(r's_waitcnt.*vmcnt', l) for l in between), name
        # the generated tap fold: eleven loads go out before its first counted wait
        fold = [n for n, l in enumerate(lines) if 'v_fmac_f32' in l and 'v252' in l]
        assert fold, name
        pre = lines[:fold[0]]
        last_asm = max(n for n, l in enumerate(pre) if 'ASMSTART' in l)
        assert sum(1 for l in pre[last_asm:] if re.search(r'\bglobal_load_dwordx4\b', l)) == 11, name
        # <1, 1, APRIO = 0 / 1 / 2>: the type-A row loop keeps ONE priority (1, its type-B partner's, in the form the launcher picks beyond
        # the Infinity Cache); <..., -1>: kernel 16's alternation 2, 2, 0.  s_setprio 1 also opens every type-B row loop.
        m = re.search(r'Lb1ELb1ELi(n?\d+)E', name)
        if m and m.group(1) in ('0', '1', '2'):
            assert not any(re.search(r'\bs_setprio 2\b', l) for l in lines) or m.group(1) == '2', name
            want_prio = int(m.group(1))
            assert sum(1 for l in lines if re.search(r'\bs_setprio %d\b' % want_prio, l)) >= 1, name
        else:
            assert sum(1 for l in lines if re.search(r'\bs_setprio 2\b', l)) >= 4, name
    assert found == 7              # <NT, REV, APRIO> = <0, 0>, <0, 1>, <1, 0>, <1, 1> at -1 and <1, 1> at 0, 1, 2


@pytest.mark.parametrize('epi', [1, 2])
def test_second_output_epilogue_loads_go_out_before_the_stores(device_asm, epi):
    """The Winograd kernel's second-output epilogues (EPI 1 / 2) load the 16 values to add ahead of the inverse transform; inside
    the store loop each was waited for where it was used (25-31 % of the wave cycles waiting: profiles/r03_wino_conv_pmc.txt)."""
    found = 0
    for tall in (0, 1):
        for name, lines in _kernel_bodies(device_asm, '_ZN4wino7conv3x3ILi0ELi0ELi0ELi0ELb%dELi%dEE' % (tall, epi)):
            found += 1
            mf = [n for n, l in enumerate(lines) if 'v_mfma' in l]
            ep = lines[mf[-1] + 1:]
            stores = [n for n, l in enumerate(ep) if re.search(r'\bbuffer_store_dwordx2\b', l)]
            loads = [n for n, l in enumerate(ep) if re.search(r'\bbuffer_load_dword\b', l)]
            assert len(loads) == 16 and stores, (name, len(loads))          # the 16 add values (the bias entered through accumulator block 5)
            assert max(loads) < min(stores), name                            # every load is issued before the first output store
    assert found == 2


@pytest.mark.parametrize('edge,nl', [(0, 6), (1, 12)])
@pytest.mark.parametrize('parts', [0, 1])
def test_split_kernel_counted_waits_registers_and_mfma_count(device_asm, parts, edge, nl):
    """The split-bf16 kernel (csrc/wino_split.hip.inc, conv3x3<ACT 1, PARTS, EPI 0, EDGE>): its weight DMA is awaited by a counted
    s_waitcnt vmcnt(NL) in front of the barrier of position 7 (and of the prologue), which assumes exactly NL compiler-tracked
    patch loads behind the last DMA; two waves per SIMD, so 128 + 128 registers and not one spill; 24 MFMAs per chunk body."""
    prefix = '_ZN4wino5split7conv3x3ILi1ELi%dELi0ELb%dELi0ELi0EE' % (parts, edge)
    found = 0
    for name, lines in _kernel_bodies(device_asm, prefix):
        found += 1
        assert not any(re.search(r'\bscratch_(load|store)', l) for l in lines), name
        assert sum(1 for l in lines if 'v_mfma_f32_32x32x16_bf16' in l) == 4 * 24, name     # chunk 0, the two loop bodies, the tail
        assert not any('v_mfma_f32_32x32x2_f32' in l for l in lines), name
        assert sum(1 for l in lines if 'global_load_lds_dwordx4' in l) == 6 + 4 * 6, name
        behind, checked = None, 0
        for n, l in enumerate(lines):
            if 'global_load_lds_dwordx4' in l:
                behind = 0
            elif behind is not None and re.search(r'\bbuffer_load_dword', l):
                behind += 1
            m = re.search(r's_waitcnt vmcnt\((\d+)\)', l)
            if m and behind is not None and any('s_barrier' in x for x in lines[n:n + 4]) and int(m.group(1)) > 0:
                # correctness: the wait may leave in flight at most what was issued BEHIND the last DMA; the prologue's DMA is its
                # oldest operation (patches of three chunks and the bias follow: 3 NL + 4 loads), the loop's wait is exact
                if checked == 0:
                    assert int(m.group(1)) == nl and behind == 3 * nl + 4, (name, n, m.group(1), behind)
                elif checked == 4:
                    # the trailing chunk: its patch loads (for a chunk that does not exist) are dead code and gone, its DMA fills a
                    # stage nobody reads: the wait is vacuous there (the loop's end waits for vmcnt(0) before LDS is reused)
                    assert int(m.group(1)) == nl and behind in (0, nl), (name, n, m.group(1), behind)
                else:
                    assert int(m.group(1)) == behind == nl, (name, n, m.group(1), behind)
                checked += 1
                behind = None
        assert checked == 5, (name, checked)                      # prologue + four chunk bodies
    assert found == 1
    text = device_asm[device_asm.find(prefix):]
    meta = text[text.find('.amdhsa_kernel'):text.find('.end_amdhsa_kernel')]
    assert re.search(r'\.amdhsa_accum_offset\s+128', meta) and re.search(r'\.amdhsa_next_free_vgpr\s+256', meta), meta[:400]


def test_build_time_invariants_hold_and_catch_a_broken_compile(device_asm):
    """video-frame-inpainting_amd/_isa_check.py (run by _native.build on the assembly of the compile that produces the library): clean on the
    real assembly; a scratch access in the generated-loop kernel and a missing MFMA are each reported."""
    from video_frame_inpainting_amd import _isa_check
    assert _isa_check.check(device_asm) == []
    gen = next(iter(_isa_check.kernels(device_asm, '_ZN6wino4311conv3x3_gen')))
    i = device_asm.find(gen + ':')
    j = device_asm.find('v_mfma_f32_16x16x4_f32', i)
    doctored = device_asm[:j] + 'scratch_store_dword off, v5, off\n\t' + device_asm[j:]
    assert any('scratch' in v for v in _isa_check.check(doctored))
    k = device_asm.find('\n', j)
    assert any('MFMAs' in v for v in _isa_check.check(device_asm[:j] + 's_nop 0' + device_asm[k:]))
