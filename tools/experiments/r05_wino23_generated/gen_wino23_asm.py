#!/usr/bin/env python3
"""Generates video-frame-inpainting_amd/csrc/wino23_chunkloop.inc: the channel-chunk loop of the Winograd F(2x2, 3x3) kernel in its
generated eight-wave form (csrc/wino23_conv.hip.inc, conv3x3_gen23) as ONE inline-asm block with a fixed register map -- the
skeleton of tools/gen_wino43_asm.py (roles, double-buffered LDS stages, one barrier per chunk of 4 input channels, operands
requested two position groups ahead, the next chunk's loads / DMA between the MFMA groups) with the 16 positions of the 4 x 4
transform domain.  The layers that must keep F(2x2, 3x3)'s rounding (MC-Net's layers with fewer than 128 channels on either side:
profiles/r05_wino_f43_policy_study.txt) are a third of the configs[1] forward.

  workgroup = 64 output channels x 64 tiles (2 x 2 output pixels each) on eight waves, two per SIMD; wave (wm 0..1, wn 0..3) owns
  32 channels x 16 tiles x 16 positions = 128 accumulators a[0:127]: every B operand (transformed patches) feeds two MFMAs;
  waves 0-3: patches (thread = one 4 x 4 patch: tile t % 64, input channel t / 64 of the chunk; 12 loads, 32 vector instructions of
  B^T d B, four 16-byte LDS writes); waves 4-7: the chunk's 16 KB of transformed weights by LDS-DMA (four 1 KB runs each).

Register map -- VGPR:
    v[0:11] v[12:23] v[24:35]   three operand buffers: A block 0 (4), A block 1 (4), B (4)
    v[36+2r : 37+2r]            patch row r, columns 1, 2       v[44+r] column 0      v[48+r] column 3          (r = 0..3)
    v[52:67]                    INPUT  off_mid[4], off_left[4], off_right[4], A read base, B read base, LDS write base, 16 * lane
    v68 v69 v70                 current A / B read address, current write address;  v71 v72 v73 their (stage 0 + stage 1) sums
    v[74:89]                    row-pass results R[r][c];  the column pass writes V[i][0..3] over v[36+4i : 39+4i]
SGPR: as tools/gen_wino43_asm.py (inputs s[48:63]: part bases, part bytes, chunks per part, chunks, bytes per chunk, weight pointer, role).
Outputs: a[0:127]: position p (= 4 i + j), channel block b, register q  ->  a[(2 p + b) * 4 + q].
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_wino43_asm import Emitter, quad, S_IN, S_PART, S_PART_BYTES, S_CPP, S_NCHUNKS, S_STEP, S_ULO, S_UHI, S_ROLE, S_DESC, S_SOFF, S_LEFT, \
    S_REM, S_WP, S_DST_OTHER, S_DST_THIS, S_T, S_NEXT

KC, TM, TN = 4, 64, 64
U_STAGE_B = 4 * KC * TM * 16             # [4 position groups][4 k][64 channels][4 positions] floats: 16,384
V_STAGE_B = 4 * KC * TN * 16             # 16,384
STAGE_B = U_STAGE_B + V_STAGE_B          # 32,768
GROUP_B = KC * TM * 16                   # bytes from one position group to the next (U and V alike): 4,096

BUF = lambda i: 12 * i                   # operand buffer i: A0 at +0, A1 at +4, B at +8
MID = lambda r: 36 + 2 * r
E0 = lambda r: 44 + r
E3 = lambda r: 48 + r
V_IN = 52
OFF_MID = lambda r: V_IN + r
OFF_LEFT = lambda r: V_IN + 4 + r
OFF_RIGHT = lambda r: V_IN + 8 + r
IN_ABASE, IN_BBASE, IN_WBASE, IN_LANE16 = V_IN + 12, V_IN + 13, V_IN + 14, V_IN + 15
V_RA, V_RB, V_W = 68, 69, 70
V_SA, V_SB, V_SW = 71, 72, 73
ROWP = lambda r, c: 74 + 4 * r + c
V_LAST = 89


def emit_read(e, g, buf):
    b = BUF(buf)
    e('ds_read_b128 %s, v%d offset:%d', quad(b), V_RA, g * GROUP_B)
    e('ds_read_b128 %s, v%d offset:%d', quad(b + 4), V_RA, g * GROUP_B + 256)      # the wave's second 16-channel block
    e('ds_read_b128 %s, v%d offset:%d', quad(b + 8), V_RB, g * GROUP_B)


def emit_mfma_phase(e, extra=None):
    """64... 32 MFMAs on the current stage: 4 position groups x 4 positions x 2 channel blocks.  Groups 0 and 1 were requested just before."""
    for g in range(4):
        if g < 2:
            emit_read(e, g + 2, (g + 2) % 3)
        if extra:
            extra(g)
        e('s_waitcnt lgkmcnt(%d)', 6 if g < 2 else (3 if g == 2 else 0))
        b = BUF(g % 3)
        for j in range(4):
            p = 4 * g + j
            for blk in range(2):
                a = (2 * p + blk) * 4
                e('v_mfma_f32_16x16x4_f32 a[%d:%d], v%d, v%d, a[%d:%d]', a, a + 3, b + 4 * blk + j, b + 8 + j, a, a + 3)


def emit_patch_row_load(e, r):
    d = 's[%d:%d], s%d offen' % (S_DESC, S_DESC + 3, S_SOFF)
    e('buffer_load_dwordx2 v[%d:%d], v%d, %s', MID(r), MID(r) + 1, OFF_MID(r), d)
    e('buffer_load_dword v%d, v%d, %s', E0(r), OFF_LEFT(r), d)
    e('buffer_load_dword v%d, v%d, %s', E3(r), OFF_RIGHT(r), d)


def emit_patch_advance(e, tag):
    e('s_add_u32 s%d, s%d, s%d', S_SOFF, S_SOFF, S_STEP)
    e('s_sub_u32 s%d, s%d, 1', S_LEFT, S_LEFT)
    e('s_cmp_lg_u32 s%d, 0', S_LEFT)
    e('s_cbranch_scc1 %s', e.ref('SAMEPART_' + tag))
    e('s_mov_b32 s%d, s%d', S_DESC, S_NEXT)
    e('s_and_b32 s%d, s%d, 0xffff', S_DESC + 1, S_NEXT + 1)
    e('s_mov_b64 s[%d:%d], s[%d:%d]', S_NEXT, S_NEXT + 1, S_NEXT + 2, S_NEXT + 3)
    e('s_mov_b64 s[%d:%d], s[%d:%d]', S_NEXT + 2, S_NEXT + 3, S_NEXT + 4, S_NEXT + 5)
    e('s_mov_b32 s%d, 0', S_SOFF)
    e('s_mov_b32 s%d, s%d', S_LEFT, S_CPP)
    e.label('SAMEPART_' + tag)


def emit_patch_loads(e, tag):
    for r in range(4):
        emit_patch_row_load(e, r)
    emit_patch_advance(e, tag)


def bt(e, a, b, c, d, out):
    """B^T of F(2x2, 3x3) applied to (a, b, c, d): (a - c, b + c, c - b, b - d)"""
    e('v_sub_f32 v%d, v%d, v%d', out[0], a, c)
    e('v_add_f32 v%d, v%d, v%d', out[1], b, c)
    e('v_sub_f32 v%d, v%d, v%d', out[2], c, b)
    e('v_sub_f32 v%d, v%d, v%d', out[3], b, d)


def emit_transform(e):
    """B^T d B of the thread's patch: R[r][:] = B^T applied to row r (row pass), V[:][c] = B^T applied to column c of R; V[i][0..3] are
    positions 4 i .. 4 i + 3 = position group i: one 16-byte LDS write each."""
    for r in range(4):
        bt(e, E0(r), MID(r), MID(r) + 1, E3(r), [ROWP(r, c) for c in range(4)])
    for c in range(4):
        bt(e, ROWP(0, c), ROWP(1, c), ROWP(2, c), ROWP(3, c), [36 + 4 * i + c for i in range(4)])
    for i in range(4):
        e('ds_write_b128 v%d, %s offset:%d', V_W, quad(36 + 4 * i), i * GROUP_B)


def emit_dma_piece(e, r, first_dst=None):
    if r == 0:
        e('s_mov_b32 m0, s%d', S_DST_OTHER if first_dst is None else first_dst)
    else:
        e('s_add_u32 m0, m0, 0x1000')
    e('s_nop 0')
    e('global_load_lds_dwordx4 v%d, s[%d:%d]', IN_LANE16, S_WP, S_WP + 1)
    e('s_add_u32 s%d, s%d, 0x1000', S_WP, S_WP)
    e('s_addc_u32 s%d, s%d, 0', S_WP + 1, S_WP + 1)


def emit_toggle(e, patch):
    e('v_sub_u32 v%d, v%d, v%d', V_RA, V_SA, V_RA)
    e('v_sub_u32 v%d, v%d, v%d', V_RB, V_SB, V_RB)
    if patch:
        e('v_sub_u32 v%d, v%d, v%d', V_W, V_SW, V_W)
    else:
        e('s_mov_b32 s%d, s%d', S_T, S_DST_OTHER)
        e('s_mov_b32 s%d, s%d', S_DST_OTHER, S_DST_THIS)
        e('s_mov_b32 s%d, s%d', S_DST_THIS, S_T)


def emit_role(e, patch):
    tag = 'P' if patch else 'D'
    e.label('ROLE_' + tag)
    if patch:
        e('s_setprio 2')
        emit_patch_loads(e, tag + 'P0')
        e('s_waitcnt vmcnt(0)')
        emit_transform(e)
        e('s_cmp_lt_u32 s%d, 2', S_REM)
        e('s_cbranch_scc1 %s', e.ref('PRO_DONE_' + tag))
        emit_patch_loads(e, tag + 'P1')
        e.label('PRO_DONE_' + tag)
        e('v_sub_u32 v%d, v%d, v%d', V_W, V_SW, V_W)
        e('s_waitcnt lgkmcnt(0)')
    else:
        for r in range(4):
            emit_dma_piece(e, r, first_dst=S_DST_THIS)
        e('s_waitcnt vmcnt(0)')
    e('s_barrier')
    e.label('LOOP_' + tag)
    if patch:
        e('s_cmp_lt_u32 s%d, 2', S_REM)
        e('s_cbranch_scc1 %s', e.ref('PLAIN_' + tag))
        e('s_waitcnt vmcnt(0)')
        emit_transform(e)
        emit_read(e, 0, 0)
        emit_read(e, 1, 1)
        e('s_cmp_lt_u32 s%d, 3', S_REM)
        e('s_cbranch_scc1 %s', e.ref('PLAIN_NOREAD_' + tag))
        emit_mfma_phase(e, lambda g: emit_patch_row_load(e, g))          # one patch row (three loads) in front of each group
        emit_patch_advance(e, tag + 'L')
        e('s_branch %s', e.ref('CHUNK_END_' + tag))
        e.label('PLAIN_' + tag)
        emit_read(e, 0, 0)
        emit_read(e, 1, 1)
        e.label('PLAIN_NOREAD_' + tag)
        emit_mfma_phase(e)
        e.label('CHUNK_END_' + tag)
        e('s_waitcnt lgkmcnt(0)')
    else:
        emit_read(e, 0, 0)
        emit_read(e, 1, 1)
        e('s_cmp_lt_u32 s%d, 2', S_REM)
        e('s_cbranch_scc1 %s', e.ref('PLAIN_' + tag))
        pieces = {0: (0, 1), 1: (2, 3)}                                    # the four DMAs in front of groups 0 and 1
        emit_mfma_phase(e, lambda g: [emit_dma_piece(e, r) for r in pieces.get(g, ())])
        e('s_branch %s', e.ref('CHUNK_END_' + tag))
        e.label('PLAIN_' + tag)
        emit_mfma_phase(e)
        e.label('CHUNK_END_' + tag)
        e('s_waitcnt vmcnt(0)')
    e('s_barrier')
    emit_toggle(e, patch)
    e('s_sub_u32 s%d, s%d, 1', S_REM, S_REM)
    e('s_cmp_lg_u32 s%d, 0', S_REM)
    e('s_cbranch_scc1 %s', e.ref('LOOP_' + tag))
    e('s_branch %s', e.ref('END'))


def generate():
    e = Emitter()
    e('s_nop 4')
    for i in range(128):
        e('v_accvgpr_write_b32 a%d, 0', i)
    e('v_mov_b32 v%d, v%d', V_RA, IN_ABASE)
    e('v_mov_b32 v%d, v%d', V_RB, IN_BBASE)
    e('v_mov_b32 v%d, v%d', V_W, IN_WBASE)
    for dst, src in ((V_SA, IN_ABASE), (V_SB, IN_BBASE), (V_SW, IN_WBASE)):
        e('v_lshlrev_b32 v%d, 1, v%d', dst, src)
        e('v_add_u32 v%d, 0x%x, v%d', dst, STAGE_B, dst)
    e('s_mov_b32 s%d, s%d', S_DESC, S_PART(0)[0])
    e('s_and_b32 s%d, s%d, 0xffff', S_DESC + 1, S_PART(0)[1])
    e('s_mov_b32 s%d, s%d', S_DESC + 2, S_PART_BYTES)
    e('s_mov_b32 s%d, 0x00020000', S_DESC + 3)
    for i in range(3):
        e('s_mov_b64 s[%d:%d], s[%d:%d]', S_NEXT + 2 * i, S_NEXT + 2 * i + 1, S_PART(i + 1)[0], S_PART(i + 1)[1])
    e('s_mov_b32 s%d, 0', S_SOFF)
    e('s_mov_b32 s%d, s%d', S_LEFT, S_CPP)
    e('s_mov_b32 s%d, s%d', S_REM, S_NCHUNKS)
    e('s_mov_b64 s[%d:%d], s[%d:%d]', S_WP, S_WP + 1, S_ULO, S_UHI)
    e('s_lshr_b32 s%d, s%d, 8', S_T, S_ROLE)
    e('s_lshl_b32 s%d, s%d, 10', S_DST_THIS, S_T)
    e('s_add_u32 s%d, s%d, %d', S_DST_OTHER, S_DST_THIS, STAGE_B)
    e('s_and_b32 s%d, s%d, 0xff', S_T, S_ROLE)
    e('s_cmp_eq_u32 s%d, 0', S_T)
    e('s_cbranch_scc1 %s', e.ref('ROLE_P'))
    e('s_branch %s', e.ref('ROLE_D'))
    emit_role(e, True)
    emit_role(e, False)
    e.label('END')
    e('s_setprio 0')
    e('s_nop 15')
    e('s_nop 15')
    return e.lines


def generate_inverse(blk):
    """A^T m A of channel block blk, registers r = 0..3: m[i][j] = a[(2 (4 i + j) + blk) * 4 + r]  ->  v[4 r + 2 a + b] = output (a, b) of the
    2 x 2 tile.  A^T = [[1, 1, 1, 0], [0, 1, -1, -1]].  Temporaries v[16:31]."""
    e = Emitter()
    M = lambda i: 16 + i                 # a column of m
    T = lambda a, j: 20 + 4 * a + j      # t[a][j], a = 0..1
    for r in range(4):
        for j in range(4):
            for i in range(4):
                e('v_accvgpr_read_b32 v%d, a%d', M(i), (2 * (4 * i + j) + blk) * 4 + r)
            e('v_add_f32 v%d, v%d, v%d', T(0, j), M(0), M(1))
            e('v_sub_f32 v%d, v%d, v%d', T(1, j), M(1), M(2))
            e('v_add_f32 v%d, v%d, v%d', T(0, j), T(0, j), M(2))            # (m0 + m1) + m2
            e('v_sub_f32 v%d, v%d, v%d', T(1, j), T(1, j), M(3))            # (m1 - m2) - m3
        for a in range(2):
            o = 4 * r + 2 * a
            e('v_add_f32 v%d, v%d, v%d', o, T(a, 0), T(a, 1))
            e('v_sub_f32 v%d, v%d, v%d', o + 1, T(a, 1), T(a, 2))
            e('v_add_f32 v%d, v%d, v%d', o, o, T(a, 2))
            e('v_sub_f32 v%d, v%d, v%d', o + 1, o + 1, T(a, 3))
    return e.lines


def clobbers():
    v = ['"v%d"' % i for i in list(range(0, V_IN)) + list(range(V_IN + 16, V_LAST + 1))]
    s = ['"s%d"' % i for i in range(S_DESC, S_NEXT + 6)]
    return ', '.join(v + s + ['"vcc"', '"scc"', '"memory"'])


def main():
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'video-frame-inpainting_amd', 'csrc', 'wino23_chunkloop.inc')
    lines = generate()
    text = ['// GENERATED by tools/gen_wino23_asm.py -- do not edit.  Register map and schedule: see the generator.',
            '// %d instructions (%d MFMAs in two role loops).' % (sum(1 for l in lines if not l.endswith(':')), sum(1 for l in lines if 'v_mfma' in l)),
            '#define TAI_W23_STAGE_BYTES %d' % STAGE_B,
            '#define TAI_W23_LOOP_ASM \\']
    for l in lines:
        text.append('    "%s\\n" \\' % l)
    text.append('    ""')
    text.append('#define TAI_W23_LOOP_CLOBBERS %s' % clobbers())
    for blk in range(2):
        text.append('#define TAI_W23_INVERSE_ASM_B%d \\' % blk)
        for l in generate_inverse(blk):
            text.append('    "%s\\n" \\' % l)
        text.append('    ""')
    text.append('#define TAI_W23_INVERSE_CLOBBERS %s' % ', '.join('"v%d"' % i for i in range(16, 32)))
    tmp = out + '.tmp.%d' % os.getpid()
    with open(tmp, 'w') as f:
        f.write('\n'.join(text) + '\n')
    os.replace(tmp, out)
    print('wrote %s: %d lines' % (os.path.normpath(out), len(lines)))


if __name__ == '__main__':
    main()
