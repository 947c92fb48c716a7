#!/usr/bin/env python3
"""Generates video-frame-inpainting_amd/csrc/sepconv_fwd_rowloop.inc: the hand-scheduled gfx950 row loop of
the forward separable convolution, as one inline-asm block with a fixed register map.

Why hand-scheduled: under the 256-VGPR budget hipcc (ROCm 7.2) either sinks the v prefetch to the end
of the loop body and waits on it immediately, or serialises every ds_read behind an lgkmcnt(0)
(profiles/r01_notes.md).  Here every load, wait and FMA has a fixed slot:

  per input row (rolled loop, 51 trips for ks = 51), per lane (4 adjacent output pixels):
    * 1 LDS-DMA (global_load_lds_dwordx4) of the v plane RING rows ahead into a per-wave LDS ring
      -- no VGPRs are held by loads in flight; the wave waits with a COUNTED vmcnt(RING-1);
    * 1 ds_read_b128 of this row's v values from the ring;
    * 14 window reads (13 ds_read_b128 + 1 ds_read_b64), issued 3 chunks ahead of their use, the
      last three reaching into the next row, each consumed behind a counted lgkmcnt;
    * 100 v_pk_fma_f32 + 4 v_fmac_f32 (row sums) + 4 v_pk_fma_f32 (fold with v).

Register map (VGPR):
    v[51p + t]            horizontal tap t of pixel p (p = 0..3, t = 0..50)          v0   .. v203
    v[204+2p : 205+2p]    ACC_p: (even-slot, odd-slot) partial row sums of pixel p     v204 .. v211
    v[212+2p : 213+2p]    O_p:   packed output accumulators                            v212 .. v219
    v[220+4j : 223+4j]    window buffer j (j = chunk % 4)                               v220 .. v235
    v[236:239]            v values of the row                                           v236 .. v239
    v240 row window LDS address, v241 ring LDS address (slot 0), v242 global byte offset, v243 temp
Tap pairing (see sepconv_forward_packed): even pixels pair taps (2i, 2i+1), odd pixels (2i-1, 2i);
both multiply the aligned window pair W2[m] = (w[2m], w[2m+1]) with m = i + (p >> 1).
"""
import os
import re

KS = 51
PITCH_BYTES = 180 * 4
RING = 7                 # v rows in flight
SLOTS = RING + 1         # ring slots (power of two)
LOOKAHEAD = 3            # window chunks in flight ahead of the one being consumed
NBUF = 4                 # window buffers (LOOKAHEAD + 1)
ABLATE_VMCNT = False     # timing experiments only: drop the v-ring wait / the window waits (results are then wrong)
ABLATE_LGKM = False
PRIO_ALTERNATE = False   # type A: s_setprio 2 on even rows, 0 on odd rows (its type-B SIMD partner sits at 1)
PRIO_CONST = None        # type A at ONE priority for the whole row loop (0, 1 or 2; the *_P<n>_NT forms of the persistent kernel): with
                         # the alternation above a type-A wave catches up with its type-B partner and both then fall into step -- in
                         # their row loops together, in their memory phases together; a constant 0 lets the type-B wave (at 1) run
                         # ahead, so that one wave of a SIMD computes while the other streams its taps
NCHUNK = 14              # 13 x b128 + 1 x b64 = 27 window pairs
NW2 = 27

A = lambda p, t: 51 * p + t
ACC = lambda p: 204 + 2 * p
O = lambda p: 212 + 2 * p
BUF = lambda k: 220 + 4 * (k % NBUF)   # k = running chunk number (14 per row: the rotation shifts per row)
VV = 244              # v values of the row: v[244:247]
V_ROW_IN, V_RING, V_GOFF, V_TMP = 240, 241, 242, 243
V_ROW = 248          # running copy of the row window address (v240 itself is left untouched)

# scalar registers (all named explicitly and listed as clobbers, except the inputs)
S_VPTR_IN = 's[60:61]'   # input: &v[b, 0, 0, 0]
S_PLANE = 's62'          # input: H*W*4
S_RINGM0 = 's63'         # input: LDS byte offset of this wave's ring (slot 0)
S_PTR_LO, S_PTR_HI = 's64', 's65'   # running DMA source pointer
S_ROW = 's66'            # row counter fy
S_SLOT_RD = 's67'        # byte offset of the slot read this row   = (fy % SLOTS) * 1024
S_SLOT_WR = 's68'        # byte offset of the slot the DMA fills    = ((fy + RING) % SLOTS) * 1024
S_T0, S_T1 = 's69', 's70'


NT = ''      # ' nt' while the *_NT variants are generated: once-read tap streams of launches whose tap footprint exceeds the
             # Infinity Cache (the persistent forward kernel) load non-temporally -- 7.0-7.2 TB/s against 6.3-6.5 on a 1.2 GB sweep
             # (profiles/r04_hbm_stream_microbench.txt)


def pair(r):
    return 'v[%d:%d]' % (r, r + 1)


def emit_chunk_read(lines, k, row_off, base=0):
    """window chunk k of the row at byte offset row_off from v240; base = running number of that row's chunk 0."""
    off = row_off + 16 * k
    b = BUF(base + k)
    if k == NCHUNK - 1:
        lines.append('ds_read_b64 v[%d:%d], v%d offset:%d' % (b, b + 1, V_ROW, off))
    else:
        lines.append('ds_read_b128 v[%d:%d], v%d offset:%d' % (b, b + 3, V_ROW, off))


def emit_chunk_fmas(lines, k, first_done, base=0):
    """FMAs that consume window pairs m = 2k, 2k+1 held in buffer k."""
    ops = []
    for hh in range(2):
        m = 2 * k + hh
        if m >= NW2:
            continue
        w = BUF(base + k) + 2 * hh
        for p in range(4):
            i = m - (p >> 1)
            if i < 0 or i > 25:
                continue
            if p % 2 == 0:
                if i <= 24:
                    ops.append(('pk', p, A(p, 2 * i), w))
                else:
                    ops.append(('lo', p, A(p, 50), w))       # tap 50 alone: ACC.lo += h50 * W.lo
            else:
                if i == 0:
                    ops.append(('hi', p, A(p, 0), w))        # tap 0 alone: ACC.hi += h0 * W.hi
                else:
                    ops.append(('pk', p, A(p, 2 * i - 1), w))
    # a pixel's first instruction of the row must be the packed one (it initialises both halves)
    ops.sort(key=lambda o: 0 if (o[0] == 'pk' or o[1] in first_done) else 1)
    for kind, p, a, w in ops:
        if kind == 'pk':
            if p in first_done:
                lines.append('v_pk_fma_f32 %s, %s, %s, %s' % (pair(ACC(p)), pair(a), pair(w), pair(ACC(p))))
            else:
                lines.append('v_pk_mul_f32 %s, %s, %s' % (pair(ACC(p)), pair(a), pair(w)))
                first_done.add(p)
        elif kind == 'lo':
            assert p in first_done
            lines.append('v_fmac_f32 v%d, v%d, v%d' % (ACC(p), a, w))
        else:
            if p in first_done:
                lines.append('v_fmac_f32 v%d, v%d, v%d' % (ACC(p) + 1, a, w + 1))
            else:   # pixel 3: its lone tap 0 comes one chunk before its first pair
                lines.append('v_mul_f32 v%d, v%d, v%d' % (ACC(p) + 1, a, w + 1))
                lines.append('v_mov_b32 v%d, 0' % ACC(p))
                first_done.add(p)


PRIO_PATTERN = None      # type A: priorities of rows 3k, 3k + 1, 3k + 2 (the *_Q<abc>_NT forms; PRIO_ALTERNATE is the pattern (2, 2, 0))


def emit_row(L, phase):
    base = NCHUNK * phase            # running chunk number of this row's chunk 0
    if PRIO_ALTERNATE:
        # the type-B partner sits at priority 1: this wave wins the SIMD's VALU arbitration on 2 rows out of 3
        L.append('s_setprio %d' % (0 if phase % 3 == 2 else 2))
    elif PRIO_PATTERN is not None:
        L.append('s_setprio %d' % PRIO_PATTERN[phase % 3])
    if not ABLATE_VMCNT:
        L.append('s_waitcnt vmcnt(%d)' % (RING - 1))
    L.append('v_add_u32 v%d, %s, v%d' % (V_TMP, S_SLOT_RD, V_RING))
    L.append('ds_read_b128 v[%d:%d], v%d' % (VV, VV + 3, V_TMP))
    # DMA of row fy + RING (source pointer stops advancing at the last row: the tail re-fetches row ks-1)
    L.append('s_add_u32 %s, %s, %s' % (S_T0, S_RINGM0, S_SLOT_WR))
    L.append('s_mov_b32 m0, %s' % S_T0)
    L.append('s_nop 0')
    L.append('global_load_lds_dwordx4 v%d, s[64:65]%s' % (V_GOFF, NT))
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, KS - 1 - RING))
    L.append('s_cselect_b32 %s, %s, 0' % (S_T1, S_PLANE))
    L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_T1))
    L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    L.append('s_add_u32 %s, %s, 1024' % (S_SLOT_RD, S_SLOT_RD))
    L.append('s_and_b32 %s, %s, %d' % (S_SLOT_RD, S_SLOT_RD, SLOTS * 1024 - 1))
    L.append('s_add_u32 %s, %s, 1024' % (S_SLOT_WR, S_SLOT_WR))
    L.append('s_and_b32 %s, %s, %d' % (S_SLOT_WR, S_SLOT_WR, SLOTS * 1024 - 1))
    first_done = set()
    for k in range(NCHUNK):
        nk = k + LOOKAHEAD
        if nk < NCHUNK:
            emit_chunk_read(L, nk, 0, base)
        else:
            emit_chunk_read(L, nk - NCHUNK, PITCH_BYTES, base + NCHUNK)
        # outstanding LDS ops after this issue, oldest first: chunk k, k+1, k+2, (v row if k < 3), chunk k+3
        L.append('s_waitcnt lgkmcnt(%d)' % (15 if ABLATE_LGKM else LOOKAHEAD + (1 if k < LOOKAHEAD else 0)))
        emit_chunk_fmas(L, k, first_done, base)
    # fold with v:  O_p += (v_p, v_p) * ACC_p
    L.append('v_pk_fma_f32 %s, %s, %s, %s op_sel_hi:[0,1,1]' % (pair(O(0)), pair(VV), pair(ACC(0)), pair(O(0))))
    L.append('v_pk_fma_f32 %s, %s, %s, %s op_sel:[1,0,0] op_sel_hi:[1,1,1]' % (pair(O(1)), pair(VV), pair(ACC(1)), pair(O(1))))
    L.append('v_pk_fma_f32 %s, %s, %s, %s op_sel_hi:[0,1,1]' % (pair(O(2)), pair(VV + 2), pair(ACC(2)), pair(O(2))))
    L.append('v_pk_fma_f32 %s, %s, %s, %s op_sel:[1,0,0] op_sel_hi:[1,1,1]' % (pair(O(3)), pair(VV + 2), pair(ACC(3)), pair(O(3))))
    L.append('v_add_u32 v%d, %d, v%d' % (V_ROW, PITCH_BYTES, V_ROW))
    L.append('s_add_u32 %s, %s, 1' % (S_ROW, S_ROW))


def gen():
    L = []
    # ---------------- prologue: O = 0, start RING DMAs, start the first LOOKAHEAD window reads
    for p in range(4):
        L.append('v_mov_b32 v%d, 0' % O(p))
        L.append('v_mov_b32 v%d, 0' % (O(p) + 1))
    L.append('v_mov_b32 v%d, v%d' % (V_ROW, V_ROW_IN))
    L.append('s_mov_b32 %s, s60' % S_PTR_LO)
    L.append('s_mov_b32 %s, s61' % S_PTR_HI)
    for r in range(RING):
        L.append('s_add_u32 %s, %s, %d' % (S_T0, S_RINGM0, r * 1024))
        L.append('s_mov_b32 m0, %s' % S_T0)
        L.append('s_nop 0')
        L.append('global_load_lds_dwordx4 v%d, s[64:65]%s' % (V_GOFF, NT))
        L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_PLANE))
        L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    # s[64:65] now points at row RING
    L.append('s_mov_b32 %s, 0' % S_ROW)
    if PRIO_CONST is not None:
        L.append('s_setprio %d' % PRIO_CONST)
    L.append('s_mov_b32 %s, 0' % S_SLOT_RD)
    L.append('s_mov_b32 %s, %d' % (S_SLOT_WR, RING * 1024))
    for k in range(LOOKAHEAD):
        emit_chunk_read(L, k, 0)
    L.append('.p2align 6')   # (labels are local numeric ones so the block can be instantiated twice)
    L.append('1:')
    # 14 chunks per row over 4 buffers: the rotation shifts by 2 per row, so the loop body is TWO rows
    # (phase 0 and phase 1) and the odd 51st row is emitted once more behind the loop.
    from math import gcd
    period = NBUF // gcd(NCHUNK, NBUF)          # rows after which the buffer rotation repeats
    body_rows = period * 3 if (PRIO_ALTERNATE or PRIO_PATTERN is not None) else period
    n_loops = KS // body_rows
    for ph in range(body_rows):
        emit_row(L, ph)
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, n_loops * body_rows))
    L.append('s_cbranch_scc1 1b')
    for ph in range(KS - n_loops * body_rows):
        emit_row(L, ph)
    # ---------------- drain: the three window reads issued for the (non-existent) next row, the DMA tail
    L.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    if PRIO_ALTERNATE or PRIO_CONST is not None or PRIO_PATTERN is not None:
        L.append('s_setprio 0')
    return L


# ---------------------------------------------------------------------------------------------------------------------
# Type B row loop ("taps last"): the lane keeps ks x 4 ACCUMULATORS c_p[fx] = sum_fy v_p[fy] * in[y+fy, x+p+fx]
# register-resident (same register layout as the taps: c_p[t] in v[51p + t]), streams v one plane per row exactly like
# type A, and has NO per-row fold: 100 v_pk_fma_f32 + 4 v_fmac_f32 per row.  The taps are needed only after the loop
# (out_p = sum_fx h_p[fx] * c_p[fx], done in HIP C++), so a type-B wave starts its FMAs at once and streams h at the END
# of its life, while a type-A wave streams h at the START: one of each per SIMD keeps both HBM and the VALU busy.
VVB = (244, 236)         # v values of the current row alternate between v[244:247] and v[236:239]


def emit_chunk_fmas_b(lines, k, base, vv):
    for hh in range(2):
        m = 2 * k + hh
        if m >= NW2:
            continue
        w = BUF(base + k) + 2 * hh
        for p in range(4):
            i = m - (p >> 1)
            if i < 0 or i > 25:
                continue
            vp = vv + 2 * (p >> 1)                      # register pair holding v_p
            sel = 'op_sel_hi:[0,1,1]' if p % 2 == 0 else 'op_sel:[1,0,0] op_sel_hi:[1,1,1]'
            if p % 2 == 0:
                if i <= 24:
                    c = A(p, 2 * i)
                    lines.append('v_pk_fma_f32 %s, %s, %s, %s %s' % (pair(c), pair(vp), pair(w), pair(c), sel))
                else:
                    lines.append('v_fmac_f32 v%d, v%d, v%d' % (A(p, 50), vp, w))               # c50 += v_p * W.lo
            else:
                if i == 0:
                    lines.append('v_fmac_f32 v%d, v%d, v%d' % (A(p, 0), vp + 1, w + 1))        # c0 += v_p * W.hi
                else:
                    c = A(p, 2 * i - 1)
                    lines.append('v_pk_fma_f32 %s, %s, %s, %s %s' % (pair(c), pair(vp), pair(w), pair(c), sel))


def emit_row_b(L, phase):
    base = NCHUNK * phase
    vv_cur, vv_next = VVB[phase % 2], VVB[(phase + 1) % 2]
    L.append('s_waitcnt vmcnt(%d)' % (RING - 2))                      # row fy+1 of v has landed in the ring
    L.append('v_add_u32 v%d, %s, v%d' % (V_TMP, S_SLOT_RD, V_RING))   # S_SLOT_RD = slot of row fy+1
    L.append('ds_read_b128 v[%d:%d], v%d' % (vv_next, vv_next + 3, V_TMP))
    L.append('s_add_u32 %s, %s, %s' % (S_T0, S_RINGM0, S_SLOT_WR))
    L.append('s_mov_b32 m0, %s' % S_T0)
    L.append('s_nop 0')
    L.append('global_load_lds_dwordx4 v%d, s[64:65]%s' % (V_GOFF, NT))
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, KS - 1 - RING))
    L.append('s_cselect_b32 %s, %s, 0' % (S_T1, S_PLANE))
    L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_T1))
    L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    L.append('s_add_u32 %s, %s, 1024' % (S_SLOT_RD, S_SLOT_RD))
    L.append('s_and_b32 %s, %s, %d' % (S_SLOT_RD, S_SLOT_RD, SLOTS * 1024 - 1))
    L.append('s_add_u32 %s, %s, 1024' % (S_SLOT_WR, S_SLOT_WR))
    L.append('s_and_b32 %s, %s, %d' % (S_SLOT_WR, S_SLOT_WR, SLOTS * 1024 - 1))
    for k in range(NCHUNK):
        nk = k + LOOKAHEAD
        if nk < NCHUNK:
            emit_chunk_read(L, nk, 0, base)
        else:
            emit_chunk_read(L, nk - NCHUNK, PITCH_BYTES, base + NCHUNK)
        L.append('s_waitcnt lgkmcnt(%d)' % (LOOKAHEAD + (1 if k < LOOKAHEAD else 0)))
        emit_chunk_fmas_b(L, k, base, vv_cur)
    L.append('v_add_u32 v%d, %d, v%d' % (V_ROW, PITCH_BYTES, V_ROW))
    L.append('s_add_u32 %s, %s, 1' % (S_ROW, S_ROW))


def gen_b():
    L = []
    for r in range(0, 204, 2):
        L.append('v_mov_b64 v[%d:%d], 0' % (r, r + 1))
    L.append('v_mov_b32 v%d, v%d' % (V_ROW, V_ROW_IN))
    L.append('s_mov_b32 %s, s60' % S_PTR_LO)
    L.append('s_mov_b32 %s, s61' % S_PTR_HI)
    for r in range(RING):
        L.append('s_add_u32 %s, %s, %d' % (S_T0, S_RINGM0, r * 1024))
        L.append('s_mov_b32 m0, %s' % S_T0)
        L.append('s_nop 0')
        L.append('global_load_lds_dwordx4 v%d, s[64:65]%s' % (V_GOFF, NT))
        L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_PLANE))
        L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    L.append('s_mov_b32 %s, 0' % S_ROW)
    L.append('s_setprio 1')
    L.append('s_mov_b32 %s, 1024' % S_SLOT_RD)                         # the in-loop read fetches row fy+1
    L.append('s_mov_b32 %s, %d' % (S_SLOT_WR, RING * 1024))
    L.append('s_waitcnt vmcnt(%d)' % (RING - 1))                       # row 0 has landed
    L.append('ds_read_b128 v[%d:%d], v%d' % (VVB[0], VVB[0] + 3, V_RING))
    for k in range(LOOKAHEAD):
        emit_chunk_read(L, k, 0)
    L.append('.p2align 6')
    L.append('1:')
    assert NBUF == 4 and NCHUNK == 14
    emit_row_b(L, 0)
    emit_row_b(L, 1)
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, KS - 1))
    L.append('s_cbranch_scc1 1b')
    emit_row_b(L, 0)
    L.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    L.append('s_setprio 0')
    return L


# ---------------------------------------------------------------------------------------------------------------------
# gV row loop: the forward's type-A inner product without the v stream:  gV[fy] = gO * sum_fx h[fx] * in[y+fy, x+fx]
# (single channel).  Per row: 100 v_pk_fma_f32 + 4 v_fmac_f32 + 4 v_add_f32 + 4 v_mul_f32 and ONE 16-byte store per lane
# (s[64:65] = &gV[b, fy, 0, 0] advances by one plane per row; v242 = the lane's byte offset in the plane; lanes whose
# pixels lie outside the image are masked out of the store with s[72:73]).  gO sits in v[244:247].
def emit_row_gv(L, phase):
    base = NCHUNK * phase
    first_done = set()
    for k in range(NCHUNK):
        nk = k + LOOKAHEAD
        if nk < NCHUNK:
            emit_chunk_read(L, nk, 0, base)
        else:
            emit_chunk_read(L, nk - NCHUNK, PITCH_BYTES, base + NCHUNK)
        L.append('s_waitcnt lgkmcnt(%d)' % LOOKAHEAD)
        emit_chunk_fmas(L, k, first_done, base)
    for p in range(4):
        L.append('v_add_f32 v%d, v%d, v%d' % (236 + p, ACC(p), ACC(p) + 1))
    for p in range(4):
        L.append('v_mul_f32 v%d, v%d, v%d' % (236 + p, 244 + p, 236 + p))
    L.append('s_and_b64 exec, exec, s[72:73]')
    L.append('global_store_dwordx4 v%d, v[236:239], s[64:65]' % V_GOFF)
    L.append('s_mov_b64 exec, s[74:75]')
    L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_PLANE))
    L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    L.append('v_add_u32 v%d, %d, v%d' % (V_ROW, PITCH_BYTES, V_ROW))
    L.append('s_add_u32 %s, %s, 1' % (S_ROW, S_ROW))


# ---------------------------------------------------------------------------------------------------------------------
# type-B epilogue of the persistent forward kernel: out_p = sum_t h_p[t] * c_p[t] with the accumulators c in v[0:203] (the
# row loop's outputs, tap t of pixel p in v[51 p + t]) and h streamed once, 16 bytes per lane per tap, ELEVEN loads in flight
# (v[204:239] and v[244:251]; the HIP C++ form of this fold, compiled inside the tile loop, was left with one free register quad and waited
# for every load before issuing the next).  s[60:61] = &h[b, 0, 0, 0], s62 = plane bytes, v242 = the lane's byte offset in the
# plane; result in v[252:255].
FOLD_BASES = [204 + 4 * k for k in range(9)] + [244, 248]     # eleven register quads (tuples start on even registers); v240-v242 are inputs and stay intact
FOLD_BUFS = len(FOLD_BASES)


def gen_fold():
    L = []
    L.append('s_mov_b32 %s, s60' % S_PTR_LO)
    L.append('s_mov_b32 %s, s61' % S_PTR_HI)
    for p in range(4):
        L.append('v_mov_b32 v%d, 0' % (252 + p))

    def issue(t):
        base = FOLD_BASES[t % FOLD_BUFS]
        L.append('global_load_dwordx4 v[%d:%d], v%d, s[64:65]%s' % (base, base + 3, V_GOFF, NT))
        L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_PLANE))
        L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    for t in range(FOLD_BUFS):
        issue(t)
    for t in range(KS):
        # loads t .. min(t + FOLD_BUFS, KS) - 1 are outstanding; load t is the oldest
        L.append('s_waitcnt vmcnt(%d)' % (min(t + FOLD_BUFS, KS) - 1 - t))
        base = FOLD_BASES[t % FOLD_BUFS]
        for p in range(4):
            L.append('v_fmac_f32 v%d, v%d, v%d' % (252 + p, base + p, KS * p + t))
        if t + FOLD_BUFS < KS:
            issue(t + FOLD_BUFS)
    return L


GV_PRIO = None          # gV waves at a constant priority (their gH partners, the forward's type B, sit at 1); None: left at 0


def gen_gv():
    L = []
    if GV_PRIO is not None:
        L.append('s_setprio %d' % GV_PRIO)
    L.append('v_mov_b32 v%d, v%d' % (V_ROW, V_ROW_IN))
    L.append('s_mov_b32 %s, s60' % S_PTR_LO)
    L.append('s_mov_b32 %s, s61' % S_PTR_HI)
    L.append('s_mov_b64 s[74:75], exec')
    L.append('s_mov_b32 %s, 0' % S_ROW)
    for k in range(LOOKAHEAD):
        emit_chunk_read(L, k, 0)
    L.append('.p2align 6')
    L.append('1:')
    emit_row_gv(L, 0)
    emit_row_gv(L, 1)
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, KS - 1))
    L.append('s_cbranch_scc1 1b')
    emit_row_gv(L, 0)
    L.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    if GV_PRIO is not None:
        L.append('s_setprio 0')
    return L


# ---------------------------------------------------------------------------------------------------------------------
# Three-channel type-A row loop (RGB frames, configs[3]): the taps are channel-independent, so a lane walks the THREE channel
# patches per tap row -- 3 x (14 window reads, 100 v_pk_fma_f32 + 4 v_fmac_f32) -- and folds each channel's row sums with the
# SAME v values: v and h are read from HBM exactly once (the per-channel passes of sepconv_forward_asm_channels re-streamed
# v: measured traffic 1.94x algorithmic).  LDS: three (16 + 50)-row patches = 141 KB leave 16 KB for the v rings, i.e. two
# 1 KB slots per wave (one row of v in flight; a tap row lasts 3x as long as in the one-channel loop).
#   O_c[p] (scalar, not packed: 12 registers instead of 24):  channel 0 v[212:215], channel 1 v[216:219], channel 2 v[236:239];
#   row window addresses: v248 / v249 / v250 = v240 + c * C3_PATCH_BYTES (the offset does not fit a 16-bit immediate).
C3_PATCH_BYTES = ((16 + KS - 1) * PITCH_BYTES + 1023) & ~1023
C3_SLOTS = 2
C3_V_AT = 4           # chunk of channel 0 behind which the row's v values are read from the ring
C3_O = lambda c, p: (212, 216, 236)[c] + p
C3_ROW = lambda c: 248 + c


def emit_chunk_read_c3(lines, j, base, next_row):
    """running chunk j of a row's 42 (channel j // 14, chunk j % 14); next_row: the read belongs to the following tap row."""
    c, k = divmod(j, NCHUNK)
    off = (PITCH_BYTES if next_row else 0) + 16 * k
    b = BUF(base + j)
    if k == NCHUNK - 1:
        lines.append('ds_read_b64 v[%d:%d], v%d offset:%d' % (b, b + 1, C3_ROW(c), off))
    else:
        lines.append('ds_read_b128 v[%d:%d], v%d offset:%d' % (b, b + 3, C3_ROW(c), off))


def emit_row_c3(L, phase):
    """One tap row.  LDS operations complete in issue order, so the wait that makes operation X visible is lgkmcnt(number of
    LDS operations issued after X): `queue` mirrors the issue order (the three look-ahead reads of this row's first chunks
    were issued by the previous row)."""
    base = 3 * NCHUNK * phase
    total = 3 * NCHUNK
    queue = [('chunk', j) for j in range(LOOKAHEAD)]

    def wait_for(op):
        L.append('s_waitcnt lgkmcnt(%d)' % (15 if ABLATE_LGKM else len(queue) - 1 - queue.index(op)))

    # DMA of row fy + 1 into the other slot, at once: its occupant (row fy - 1) went to registers a row ago.  Row fy's own
    # values were requested at the start of row fy - 1 and are read from the ring only at chunk C3_V_AT of channel 0, behind
    # a vmcnt(1) that leaves the DMA just issued in flight: more than a whole tap row of lead with two slots per wave.
    L.append('s_add_u32 %s, %s, %s' % (S_T0, S_RINGM0, S_SLOT_WR))
    L.append('s_mov_b32 m0, %s' % S_T0)
    L.append('s_nop 0')
    L.append('global_load_lds_dwordx4 v%d, s[64:65]' % V_GOFF)
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, KS - 2))
    L.append('s_cselect_b32 %s, %s, 0' % (S_T1, S_PLANE))
    L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_T1))
    L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    L.append('s_xor_b32 %s, %s, 1024' % (S_SLOT_WR, S_SLOT_WR))
    for c in range(3):
        first_done = set()
        for k in range(NCHUNK):
            j = c * NCHUNK + k
            nj = j + LOOKAHEAD
            if nj < total:
                emit_chunk_read_c3(L, nj, base, False)
            else:
                emit_chunk_read_c3(L, nj - total, base + total, True)
            queue.append(('chunk', nj))
            if c == 0 and k == C3_V_AT:
                if not ABLATE_VMCNT:
                    L.append('s_waitcnt vmcnt(1)')                         # row fy of v has landed (row fy + 1 may be in flight)
                L.append('v_add_u32 v%d, %s, v%d' % (V_TMP, S_SLOT_RD, V_RING))
                L.append('ds_read_b128 v[%d:%d], v%d' % (VV, VV + 3, V_TMP))
                L.append('s_xor_b32 %s, %s, 1024' % (S_SLOT_RD, S_SLOT_RD))
                queue.append(('v', 0))
            wait_for(('chunk', j))
            # the chunk's FMAs: emit_chunk_fmas indexes buffers by base + k, so pass the channel's running base
            emit_chunk_fmas(L, k, first_done, base + c * NCHUNK)
        if c == 0:
            wait_for(('v', 0))
        # fold with v:  O_c[p] += v_p * (ACC_p.lo + ACC_p.hi)
        for pp in range(4):
            L.append('v_fmac_f32 v%d, v%d, v%d' % (C3_O(c, pp), VV + pp, ACC(pp)))
            L.append('v_fmac_f32 v%d, v%d, v%d' % (C3_O(c, pp), VV + pp, ACC(pp) + 1))
    for c in range(3):
        L.append('v_add_u32 v%d, %d, v%d' % (C3_ROW(c), PITCH_BYTES, C3_ROW(c)))
    L.append('s_add_u32 %s, %s, 1' % (S_ROW, S_ROW))


def gen_c3():
    L = []
    for c in range(3):
        for pp in range(4):
            L.append('v_mov_b32 v%d, 0' % C3_O(c, pp))
    L.append('v_mov_b32 v%d, v%d' % (C3_ROW(0), V_ROW_IN))
    L.append('v_add_u32 v%d, %d, v%d' % (C3_ROW(1), C3_PATCH_BYTES, V_ROW_IN))
    L.append('v_add_u32 v%d, %d, v%d' % (C3_ROW(2), 2 * C3_PATCH_BYTES, V_ROW_IN))
    L.append('s_mov_b32 %s, s60' % S_PTR_LO)
    L.append('s_mov_b32 %s, s61' % S_PTR_HI)
    L.append('s_mov_b32 m0, %s' % S_RINGM0)                               # row 0 of v into slot 0
    L.append('s_nop 0')
    L.append('global_load_lds_dwordx4 v%d, s[64:65]' % V_GOFF)
    L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_PLANE))
    L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
    L.append('s_mov_b32 %s, 0' % S_ROW)
    L.append('s_mov_b32 %s, 0' % S_SLOT_RD)
    L.append('s_mov_b32 %s, 1024' % S_SLOT_WR)
    for j in range(LOOKAHEAD):
        emit_chunk_read_c3(L, j, 0, False)
    L.append('.p2align 6')
    L.append('1:')
    assert NBUF == 4 and (3 * NCHUNK) % NBUF == 2                            # the buffer rotation repeats every two rows
    emit_row_c3(L, 0)
    emit_row_c3(L, 1)
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, KS - 1))
    L.append('s_cbranch_scc1 1b')
    emit_row_c3(L, 0)
    L.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    return L


# ---------------------------------------------------------------------------------------------------------------------
# grad_input row loop (csrc/sepconv_bwd.hip.inc, sepconv_grad_i_strips_asm): the forward's type-A inner product with
#   * the taps = SHEARED products S'_p[t] = (gO h[50-t])[y, X_p - (50 - t)], X_p = 4q - 2 + p, in the same register order
#     (v[51p + t]), so that out_p = sum_t S'_p[t] * w[p + t] over the window w = v[fy][y, 4q - 52 ...] -- exactly the forward's
#     pairing (emit_chunk_fmas is shared);
#   * the "patch" = the row of the v plane of this tap row, in a workgroup-shared ring of 9 plane slots filled by LDS-DMA
#     three planes (one stage) at a time, two stages ahead; one s_barrier per stage;
#   * the four row sums added into the wave-private accumulator strip (ds_read_b128 at the start of the step, add,
#     ds_write_b128 behind the owner mask), 96 bytes further per tap row.
# Inputs: v240 window address in plane slot 0; v241 accumulator address of tap row 0; v242 lane * 16 (DMA offset);
#   s[60:61] &v[b, 0, row w] (DMA source of this wave's first ring row), s62 plane bytes, s63 LDS address of that row in slot 0,
#   s[72:73] DMA lane mask, s[74:75] owner lane mask, s[78:79] != 0: the wave also copies ring row w + 8 from there (the pointer
#   stays as given: the loop advances its own copy), s77 its LDS address in slot 0.
GI_PLANE_BYTES = (64 + 10 * 152 + 64) * 4       # gi2::PLANE floats
GI_SLOTS = 9
GI_ACC_STEP = 6 * 16                              # one accumulator row of a strip: Q quads x 16 bytes
GI_T = 236                                        # v[236:239]: the accumulator row being updated
GI_WIN, GI_WIN_NEXT, GI_ACCA = 248, 249, 250      # window address of this / the next tap row, accumulator address
S_SLOT_N, S_DSLOT = 's67', 's68'                  # slot byte offsets: the next tap row's, the DMA's
S_P2LO, S_P2HI = 's80', 's81'


def emit_gi_dma_stage(L):
    """LDS-DMA of the three planes of one stage into slots S_DSLOT, +1, +2 (this wave's ring rows)."""
    L.append('s_mov_b64 s[82:83], exec')
    L.append('s_mov_b64 exec, s[72:73]')
    for k in range(3):
        L.append('s_add_u32 %s, %s, %s' % (S_T0, S_RINGM0, S_DSLOT))
        L.append('s_mov_b32 m0, %s' % S_T0)
        L.append('s_nop 0')
        L.append('global_load_lds_dwordx4 v%d, s[64:65]' % V_GOFF)
        L.append('s_add_u32 %s, %s, %s' % (S_PTR_LO, S_PTR_LO, S_PLANE))
        L.append('s_addc_u32 %s, %s, 0' % (S_PTR_HI, S_PTR_HI))
        L.append('s_cmp_eq_u64 s[78:79], 0')
        L.append('s_cbranch_scc1 2f')
        L.append('s_add_u32 %s, s77, %s' % (S_T0, S_DSLOT))
        L.append('s_mov_b32 m0, %s' % S_T0)
        L.append('s_nop 0')
        L.append('global_load_lds_dwordx4 v%d, s[80:81]' % V_GOFF)
        L.append('s_add_u32 %s, %s, %s' % (S_P2LO, S_P2LO, S_PLANE))
        L.append('s_addc_u32 %s, %s, 0' % (S_P2HI, S_P2HI))
        L.append('2:')
        L.append('s_add_u32 %s, %s, %d' % (S_DSLOT, S_DSLOT, GI_PLANE_BYTES))
        L.append('s_cmp_eq_u32 %s, %d' % (S_DSLOT, GI_SLOTS * GI_PLANE_BYTES))
        L.append('s_cselect_b32 %s, 0, %s' % (S_DSLOT, S_DSLOT))
    L.append('s_mov_b64 exec, s[82:83]')


def emit_gi_chunk_read(lines, k, next_row, base):
    b = BUF(base + k)
    reg = GI_WIN_NEXT if next_row else GI_WIN
    if k == NCHUNK - 1:
        lines.append('ds_read_b64 v[%d:%d], v%d offset:%d' % (b, b + 1, reg, 16 * k))
    else:
        lines.append('ds_read_b128 v[%d:%d], v%d offset:%d' % (b, b + 3, reg, 16 * k))


def emit_row_gi(L, phase, step_in_stage, final_stage=False):
    """One tap row.  step_in_stage 0: the stage's first row (barrier first; reads nothing ahead of it); 2: its last row (does
    not read ahead into the next stage: those planes are only known to have landed behind the next barrier)."""
    base = NCHUNK * phase
    queue = []

    def wait_for(op):
        L.append('s_waitcnt lgkmcnt(%d)' % (len(queue) - 1 - queue.index(op)))

    if step_in_stage == 0:
        # stage boundary: this wave's rows of the stage have landed (the next stage may be in flight), everyone's have, and
        # the stage before it has been consumed by every wave: its slots take the DMA of two stages ahead
        if final_stage:
            L.append('s_waitcnt vmcnt(0)')
        else:
            L.append('s_cmp_eq_u64 s[78:79], 0')
            L.append('s_cbranch_scc1 3f')
            L.append('s_waitcnt vmcnt(6)')
            L.append('s_branch 4f')
            L.append('3:')
            L.append('s_waitcnt vmcnt(3)')
            L.append('4:')
        L.append('s_barrier')
        if not final_stage:
            L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, KS - 6))             # tap rows fy + 6 .. fy + 8 exist
            L.append('s_cbranch_scc0 5f')
            emit_gi_dma_stage(L)
            L.append('5:')
        for k in range(LOOKAHEAD):                                         # nothing was read ahead across the barrier
            emit_gi_chunk_read(L, k, False, base)
            queue.append(('chunk', k))
    else:
        queue = [('chunk', k) for k in range(LOOKAHEAD)]                  # issued by the previous tap row
    # the accumulator row of this tap row: read now, used after the last chunk
    L.append('ds_read_b128 v[%d:%d], v%d' % (GI_T, GI_T + 3, GI_ACCA))
    queue.append(('acc', 0))
    # window address of the next tap row (its first chunks are read ahead from here)
    L.append('v_add_u32 v%d, %s, v%d' % (GI_WIN_NEXT, S_SLOT_N, V_ROW_IN))
    first_done = set()
    for k in range(NCHUNK):
        nk = k + LOOKAHEAD
        if nk < NCHUNK:
            emit_gi_chunk_read(L, nk, False, base)
            queue.append(('chunk', nk))
        elif step_in_stage != 2:
            emit_gi_chunk_read(L, nk - NCHUNK, True, base + NCHUNK)
            queue.append(('next', nk - NCHUNK))
        wait_for(('chunk', k))
        emit_chunk_fmas(L, k, first_done, base)
    wait_for(('acc', 0))
    for pp in range(4):
        L.append('v_add_f32 v%d, v%d, v%d' % (ACC(pp), ACC(pp), ACC(pp) + 1))
    for pp in range(4):
        L.append('v_add_f32 v%d, v%d, v%d' % (GI_T + pp, GI_T + pp, ACC(pp)))
    L.append('s_mov_b64 s[82:83], exec')
    L.append('s_mov_b64 exec, s[74:75]')
    L.append('ds_write_b128 v%d, v[%d:%d]' % (GI_ACCA, GI_T, GI_T + 3))
    L.append('s_mov_b64 exec, s[82:83]')
    L.append('v_add_u32 v%d, %d, v%d' % (GI_ACCA, GI_ACC_STEP, GI_ACCA))
    L.append('v_mov_b32 v%d, v%d' % (GI_WIN, GI_WIN_NEXT))
    # slot offset of the tap row after the next
    L.append('s_add_u32 %s, %s, %d' % (S_SLOT_N, S_SLOT_N, GI_PLANE_BYTES))
    L.append('s_cmp_eq_u32 %s, %d' % (S_SLOT_N, GI_SLOTS * GI_PLANE_BYTES))
    L.append('s_cselect_b32 %s, 0, %s' % (S_SLOT_N, S_SLOT_N))
    L.append('s_add_u32 %s, %s, 1' % (S_ROW, S_ROW))


def gen_gi():
    L = []
    L.append('s_mov_b32 %s, s60' % S_PTR_LO)
    L.append('s_mov_b32 %s, s61' % S_PTR_HI)
    L.append('s_mov_b32 %s, s78' % S_P2LO)
    L.append('s_mov_b32 %s, s79' % S_P2HI)
    L.append('s_mov_b32 %s, 0' % S_DSLOT)
    emit_gi_dma_stage(L)                                                   # stages 0 and 1: tap rows 0 .. 5
    emit_gi_dma_stage(L)
    L.append('s_mov_b32 %s, 0' % S_ROW)
    L.append('s_mov_b32 %s, %d' % (S_SLOT_N, GI_PLANE_BYTES))              # slot offset of tap row 1
    L.append('v_mov_b32 v%d, v%d' % (GI_WIN, V_ROW_IN))
    L.append('v_mov_b32 v%d, v%d' % (GI_ACCA, V_RING))
    L.append('.p2align 6')
    L.append('1:')
    # window buffers rotate with period 2 tap rows, stages with period 3: the loop body is 6 tap rows, 51 = 8 x 6 + 3
    for i in range(6):
        emit_row_gi(L, i % 2, i % 3)
    L.append('s_cmp_lt_u32 %s, %d' % (S_ROW, 48))
    L.append('s_cbranch_scc1 1b')
    for i in range(3):
        emit_row_gi(L, i % 2, i % 3, final_stage=True)
    L.append('s_waitcnt vmcnt(0) lgkmcnt(0)')
    return L


PATTERNS = ()             # measured and dropped (profiles/r04_sepconv_priority_ab.txt): '112', '122', '110', '102' -- all slower than a constant 1


def main():
    global LOOKAHEAD, NBUF, ABLATE_VMCNT, ABLATE_LGKM, PRIO_ALTERNATE, PRIO_CONST, PRIO_PATTERN, NT
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, '..', 'video-frame-inpainting_amd', 'csrc', 'sepconv_fwd_rowloop.inc')
    variants = [('TAI_FWD_ROWLOOP_ASM', 3, 4, False, False),
                ('TAI_FWD_ROWLOOP_ASM_LA4', 4, 5, False, False),          # 4 chunks ahead, 5 buffers (v[220:239])
                ('TAI_FWD_ROWLOOP_ASM_NOVM', 3, 4, True, False),          # timing experiment: no v-ring wait
                ('TAI_FWD_ROWLOOP_ASM_NOLGKM', 3, 4, False, True)]        # timing experiment: no window waits
    final, out = out, out + '.tmp.%d' % os.getpid()      # written beside the target, renamed over it at the end
    with open(out, 'w') as f:
        f.write('// GENERATED by tools/gen_fwd_asm.py -- do not edit.  Register map and schedule: see the generator.\n')
        f.write('#define TAI_FWD_ROWLOOP_RING_SLOTS %d\n' % SLOTS)
        variants.append(('TAI_FWD_ROWLOOP_ASM_PRIO', 3, 4, False, False))   # type A next to a type-B partner
        variants.append(('TAI_FWD_ROWLOOP_ASM_PRIO_NT', 3, 4, False, False))   # the same, v planes loaded non-temporally
        for pc in (0, 1, 2):                                                    # type A at one constant priority (persistent kernel, nt route)
            variants.append(('TAI_FWD_ROWLOOP_ASM_P%d_NT' % pc, 3, 4, False, False))
        for pat in PATTERNS:                                                    # ... at a priority pattern over three rows
            variants.append(('TAI_FWD_ROWLOOP_ASM_Q%s_NT' % pat, 3, 4, False, False))
        for name, la, nbuf, novm, nolgkm in variants:
            LOOKAHEAD, NBUF, ABLATE_VMCNT, ABLATE_LGKM = la, nbuf, novm, nolgkm
            PRIO_ALTERNATE = '_PRIO' in name
            mpc = re.search(r'_P(\d)_NT$', name)
            PRIO_CONST = int(mpc.group(1)) if mpc else None
            mpq = re.search(r'_Q(\d\d\d)_NT$', name)
            PRIO_PATTERN = tuple(int(c) for c in mpq.group(1)) if mpq else None
            NT = ' nt' if name.endswith('_NT') else ''
            lines = gen()
            NT = ''
            PRIO_CONST = None
            PRIO_PATTERN = None
            n_pk = sum(1 for l in lines if l.startswith('v_pk_'))
            f.write('// %s: ks=%d ring=%d slots=%d lookahead=%d buffers=%d; %d instructions, %d packed.\n'
                    % (name, KS, RING, SLOTS, LOOKAHEAD, NBUF, len(lines), n_pk))
            f.write('#define %s \\\n' % name)
            for l in lines:
                f.write('    "%s\\n" \\\n' % l)
            f.write('    ""\n')
        LOOKAHEAD, NBUF, ABLATE_VMCNT, ABLATE_LGKM, PRIO_ALTERNATE = 3, 4, False, False, False
        for suffix in ('', '_NT'):
            NT = ' nt' if suffix else ''
            lines = gen_b()
            NT = ''
            f.write('// TAI_FWD_ROWLOOP_B_ASM%s (accumulators resident, taps last): %d instructions.\n' % (suffix, len(lines)))
            f.write('#define TAI_FWD_ROWLOOP_B_ASM%s \\\n' % suffix)
            for l in lines:
                f.write('    "%s\\n" \\\n' % l)
            f.write('    ""\n')
        global GV_PRIO
        for suffix, gvp in (('', None), ('_P1', 1), ('_P2', 2)):
            GV_PRIO = gvp
            lines = gen_gv()
            GV_PRIO = None
            f.write('// TAI_GV_ROWLOOP_ASM%s (gV = gO * row sums, one store per row): %d instructions.\n' % (suffix, len(lines)))
            f.write('#define TAI_GV_ROWLOOP_ASM%s \\\n' % suffix)
            for l in lines:
                f.write('    "%s\\n" \\\n' % l)
            f.write('    ""\n')
        for suffix in ('', '_NT'):
            NT = ' nt' if suffix else ''
            lines = gen_fold()
            NT = ''
            f.write('// TAI_FWD_FOLD_ASM%s (type-B tap fold of the persistent kernel, %d loads in flight): %d instructions.\n' % (suffix, FOLD_BUFS, len(lines)))
            f.write('#define TAI_FWD_FOLD_ASM%s \\\n' % suffix)
            for l in lines:
                f.write('    "%s\\n" \\\n' % l)
            f.write('    ""\n')
        clob_fold = ['v%d' % r for b in FOLD_BASES for r in range(b, b + 4)] + ['s64', 's65', 'scc', 'memory']
        f.write('#define TAI_FWD_FOLD_CLOBBERS %s\n' % ', '.join('"%s"' % c for c in clob_fold))
        lines = gen_c3()
        n_pk = sum(1 for l in lines if l.startswith('v_pk_'))
        f.write('// TAI_FWD_ROWLOOP_C3_ASM (three channel patches per tap row, v read once): %d instructions, %d packed.\n' % (len(lines), n_pk))
        f.write('#define TAI_FWD_ROWLOOP_C3_PATCH_BYTES %d\n#define TAI_FWD_ROWLOOP_C3_RING_SLOTS %d\n' % (C3_PATCH_BYTES, C3_SLOTS))
        f.write('#define TAI_FWD_ROWLOOP_C3_ASM \\\n')
        for l in lines:
            f.write('    "%s\\n" \\\n' % l)
        f.write('    ""\n')
        # timing experiments (tools build only, results wrong): the same loop without the v-ring wait / without the window waits --
        # which of the two the three-channel kernel's 25 % of waiting cycles belong to
        for suffix, novm, nolgkm in (('_NOVM', True, False), ('_NOLGKM', False, True)):
            ABLATE_VMCNT, ABLATE_LGKM = novm, nolgkm
            lines = gen_c3()
            ABLATE_VMCNT, ABLATE_LGKM = False, False
            f.write('#define TAI_FWD_ROWLOOP_C3_ASM%s \\\n' % suffix)
            for l in lines:
                f.write('    "%s\\n" \\\n' % l)
            f.write('    ""\n')
        clob_c3 = ['v%d' % r for r in list(range(204, 212)) + list(range(220, 236)) + [243, 244, 245, 246, 247, 248, 249, 250]]
        clob_c3 += ['s%d' % r for r in range(64, 71)] + ['scc', 'memory']
        f.write('#define TAI_FWD_ROWLOOP_C3_CLOBBERS %s\n' % ', '.join('"%s"' % c for c in clob_c3))
        lines = gen_gi()
        f.write('// TAI_GI_ROWLOOP_ASM (grad_input strips: sheared taps x v-plane window, accumulate into the strip): %d instructions.\n' % len(lines))
        f.write('#define TAI_GI_ROWLOOP_ASM \\\n')
        for l in lines:
            f.write('    "%s\\n" \\\n' % l)
        f.write('    ""\n')
        clob_gi = ['v%d' % r for r in list(range(204, 212)) + list(range(220, 240)) + [248, 249, 250]]
        clob_gi += ['s%d' % r for r in (64, 65, 66, 67, 68, 69, 70, 80, 81, 82, 83)] + ['scc', 'memory']
        f.write('#define TAI_GI_ROWLOOP_CLOBBERS %s\n' % ', '.join('"%s"' % c for c in clob_gi))
        clob_gv = ['v%d' % r for r in list(range(204, 212)) + list(range(220, 240)) + [248]]
        clob_gv += ['s%d' % r for r in (64, 65, 66, 74, 75)] + ['scc', 'memory']
        f.write('#define TAI_GV_ROWLOOP_CLOBBERS %s\n' % ', '.join('"%s"' % c for c in clob_gv))
        clob_b = ['v%d' % r for r in list(range(220, 240)) + [243, 244, 245, 246, 247, 248]]
        clob_b += ['s%d' % r for r in range(64, 71)] + ['scc', 'memory']
        f.write('#define TAI_FWD_ROWLOOP_B_CLOBBERS %s\n' % ', '.join('"%s"' % c for c in clob_b))
        clob = ['v%d' % r for r in list(range(204, 212)) + list(range(220, 240)) + [243, 244, 245, 246, 247, 248]]
        clob += ['s%d' % r for r in range(64, 71)] + ['scc', 'memory']   # m0 is written too; hipcc reloads it before each of its own uses
        f.write('#define TAI_FWD_ROWLOOP_CLOBBERS %s\n' % ', '.join('"%s"' % c for c in clob))
    os.replace(out, final)
    print('wrote %s' % final)


if __name__ == '__main__':
    main()
