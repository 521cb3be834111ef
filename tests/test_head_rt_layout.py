"""CPU checks of the f32 row-tile head kernel's host logic (csrc/head_rt.h, head_rt.hip), no GPU:

* the row plan returned by the library's host-only mtr_head_row_plan: every conv_final channel
  exactly once, a joint's depth slices contiguous and in order, no softmax unit across an atom
  boundary;
* the data path the kernel's address arithmetic describes -- packed weight tiles as LDS images,
  the 1 KiB global->LDS copies with their source-side swizzle / channel-row layout, the per-lane
  fragment reads, the 16x16x4 MFMA operand and result lane maps -- replayed in numpy and compared
  with a dense GEMM.  The formulas here are written from the layout contract in head_rt.hip's header
  comment, not generated from the kernel, so a slip on either side shows up as a mismatch;
* LDS bank-conflict freedom of the fragment reads (ds_read_b128 lane groups of MI355X_MICROARCH.md).
"""
import ctypes

import numpy as np
import pytest


def row_plan(J, D):
    from metrabs_amd import _lib
    lib = _lib.load()
    nt, a = ctypes.c_int32(), ctypes.c_int32()
    assert lib.mtr_head_row_plan(J, D, ctypes.byref(nt), ctypes.byref(a), None, 0) == 0
    rows = np.zeros(nt.value * 16, np.int32)
    assert lib.mtr_head_row_plan(J, D, ctypes.byref(nt), ctypes.byref(a),
                                 rows.ctypes.data_as(ctypes.c_void_p), rows.size) == 0
    return nt.value, a.value, rows


@pytest.mark.parametrize('J,D', [(17, 8), (122, 8), (17, 72), (1, 8), (5, 8), (30, 4), (7, 5), (3, 16),
                                 (17, 1), (24, 17), (9, 32), (4, 80), (17, 3), (2, 33)])
def test_row_plan_keeps_softmax_units_inside_atoms(J, D):
    n_tiles, a, rows = row_plan(J, D)
    n_out = J * (1 + D)
    used = rows[rows >= 0]
    assert sorted(used.tolist()) == list(range(n_out)), 'every channel exactly once'
    assert n_tiles * 16 <= 2 * n_out + 32 * a, 'padding bounded (worst case: D just above a multiple of 16)'
    pos = {int(c): r for r, c in enumerate(rows) if c >= 0}
    atom_rows = 16 * a
    for j in range(J):
        r0 = pos[J + j]  # depth slice 0 of joint j (channel J + d*J + j)
        for d in range(D):
            assert pos[J + d * J + j] == r0 + d, 'depth slices contiguous, in order'
        assert r0 // atom_rows == (r0 + D - 1) // atom_rows, 'a 3D unit never straddles an atom'
    if (J, D) == (17, 8):
        assert n_tiles == 10 and a == 1   # 160 rows for 153 channels
    if (J, D) == (17, 72):
        assert n_tiles == 85 and a == 5   # 17 atoms of [72 slices | 8 2D rows]


def test_shapes_outside_the_plan_are_rejected():
    from metrabs_amd import _lib
    lib = _lib.load()
    nt, a = ctypes.c_int32(), ctypes.c_int32()
    assert lib.mtr_head_row_plan(17, 81, ctypes.byref(nt), ctypes.byref(a), None, 0) == -2
    assert lib.mtr_head_row_plan(0, 8, ctypes.byref(nt), ctypes.byref(a), None, 0) == -2
    assert lib.mtr_head_row_plan(17, 8, None, None, None, 0) == -1
    small = np.zeros(16, np.int32)
    assert lib.mtr_head_row_plan(17, 8, ctypes.byref(nt), ctypes.byref(a),
                                 small.ctypes.data_as(ctypes.c_void_p), small.size) == -5
    assert lib.mtr_head_packed_bytes(1280, 17, 72, 0) > 0      # D = 72: row-tile section only
    # 16-bit, 73-row joints: no joint-group blob, the 16-bit row-tile section alone (20 stages of 64 channels)
    assert lib.mtr_head_packed_bytes(1280, 17, 72, 1) == 20 * 85 * 2048 + 85 * 16 * 8
    assert lib.mtr_head_packed_bytes(1288, 17, 72, 1) == 0     # ... which needs C % 64 == 0
    # (the 16-bit blob holds two sections since round 3: joint groups + row tiles)
    assert lib.mtr_head_packed_bytes(1280, 17, 8, 0) > 0 and lib.mtr_head_packed_bytes(1280, 17, 8, 1) > 0


# ---------------------------------------------------------------------------------------------
# replay of the kernel's data path

def swz(row):
    return (row >> 1) & 7


def pack_weights(w, rows, n_tiles, C):
    """[stage][tile][16 rows][8 slots'][4] f32: slot' of row r holds channels 4*(slot' ^ swz(r)) .. +3"""
    n_stages = (C + 31) // 32
    wt = np.zeros((n_stages, n_tiles, 16, 8, 4), np.float32)
    for st in range(n_stages):
        for t in range(n_tiles):
            for r in range(16):
                ch = rows[t * 16 + r]
                if ch < 0:
                    continue
                for sp in range(8):
                    c0 = st * 32 + 4 * (sp ^ swz(r))
                    for e in range(4):
                        if c0 + e < C:
                            wt[st, t, r, sp, e] = w[ch, c0 + e]
    return wt


def stage_lds_image(wt, feat, nhwc, st, t0, RT, cb, C, HW):
    """What the 2 RT + 8 copies of one stage leave in the LDS buffer (bytes as float32 words)."""
    chunk = 1088 // 4
    size = RT * 512 + (8 * 256 if nhwc else 8 * chunk)
    lds = np.full(size, np.nan, np.float32)
    for j in range(2 * RT + 8):
        for lane in range(64):
            dst = None
            if j < 2 * RT:  # weight tiles: linear copy of the packed image
                src = wt[st].reshape(-1)[(t0 * 512 + j * 256 + lane * 4):][:4]
                dst = j * 256 + lane * 4
            else:
                jb = j - 2 * RT
                if nhwc:  # feat [HW][C]
                    pos, slotp = jb * 8 + (lane >> 3), lane & 7
                    slot = slotp ^ swz(pos)
                    P = cb * 64 + pos
                    P = P if P < HW else 0
                    c = st * 32 + slot * 4
                    c = c if c < C else st * 32
                    src = feat[P, c:c + 4]
                    dst = RT * 512 + jb * 256 + lane * 4
                else:  # feat [C][HW]
                    ch, p = jb * 4 + (lane >> 4), cb * 64 + (lane & 15) * 4
                    p = p if p < HW else 0
                    c = st * 32 + ch
                    c = c if c < C else st * 32
                    src = feat[c, p:p + 4]
                    dst = RT * 512 + jb * chunk + lane * 4
            lds[dst:dst + 4] = src
    return lds


def replay_block(w, feat_chw, rows, n_tiles, t0, RT, nhwc, cb):
    """logits [RT*16, 64] of one (block, column block) as the kernel's lanes would produce them."""
    C, HW = feat_chw.shape
    feat = np.ascontiguousarray(feat_chw.T) if nhwc else feat_chw
    wt = pack_weights(w, rows, n_tiles, C)
    n_stages = (C + 31) // 32
    acc = np.zeros((4, RT, 64, 4), np.float64)  # [wave][tile][lane][reg]
    for st in range(n_stages):
        lds = stage_lds_image(wt, feat, nhwc, st, t0, RT, cb, C, HW)
        # MFMA per wave / tile / k-step: D[row][col] += sum_kk A[row][kk] * B[kk][col], lane (i16, g4)
        # supplies A[i16][g4] and B[g4][i16] and owns D[4 g4 + r][i16]
        for wave in range(4):
            for q in range(2):
                A = np.zeros((RT, 4, 16, 4), np.float64)  # [tile][k-step][row][kk]
                Bm = np.zeros((4, 4, 16), np.float64)     # [k-step][kk][col]
                for lane in range(64):
                    i16, g4 = lane & 15, lane >> 4
                    a_addr = (i16 * 128 + ((g4 ^ swz(i16)) << 4)) ^ (64 * q)
                    if nhwc:
                        pos = wave * 16 + i16
                        b_addr = (pos * 128 + ((g4 ^ swz(pos)) << 4)) ^ (64 * q)
                        fb = lds[RT * 512 + b_addr // 4:][:4]
                    else:
                        b_addr = g4 * 1088 + (wave * 16 + i16) * 4 + q * 4 * 1088
                        fb = lds[RT * 512 + b_addr // 4 + np.array([0, 64, 128, 192])]
                    for k in range(4):
                        Bm[k, g4, i16] = fb[k]
                    for t in range(RT):
                        fa = lds[t * 512 + a_addr // 4:][:4]
                        for k in range(4):
                            A[t, k, i16, g4] = fa[k]
                for t in range(RT):
                    for k in range(4):
                        Dt = A[t, k] @ Bm[k]  # [row][col]
                        for lane in range(64):
                            i16, g4 = lane & 15, lane >> 4
                            for r in range(4):
                                acc[wave, t, lane, r] += Dt[4 * g4 + r, i16]
    logits = np.zeros((RT * 16, 64))
    for wave in range(4):
        for t in range(RT):
            for lane in range(64):
                i16, g4 = lane & 15, lane >> 4
                for r in range(4):
                    logits[t * 16 + 4 * g4 + r, wave * 16 + i16] = acc[wave, t, lane, r]
    return logits


@pytest.mark.parametrize('nhwc', [False, True])
@pytest.mark.parametrize('J,D,C,HW,t0,RT,cb', [(17, 8, 40, 64, 6, 3, 0), (17, 8, 64, 36, 9, 1, 0),
                                               (5, 8, 32, 144, 0, 2, 2), (3, 20, 36, 16, 2, 2, 0)])
def test_replayed_data_path_equals_dense_gemm(J, D, C, HW, t0, RT, cb, nhwc):
    rng = np.random.default_rng(J * 1000 + D * 10 + C)
    n_tiles, a, rows = row_plan(J, D)
    assert t0 + RT <= n_tiles
    w = rng.standard_normal((J * (1 + D), C)).astype(np.float32)
    feat = rng.standard_normal((C, HW)).astype(np.float32)
    got = replay_block(w, feat, rows, n_tiles, t0, RT, nhwc, cb)
    dense = w.astype(np.float64) @ feat.astype(np.float64)
    for r in range(RT * 16):
        ch = rows[t0 * 16 + r]
        for p in range(64):
            P = cb * 64 + p
            if P >= HW:
                continue  # padded columns hold finite garbage the decode never reads
            want = dense[ch, P] if ch >= 0 else 0.0
            assert abs(got[r, p] - want) < 1e-9, (r, p, ch)


B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[x + 32 for x in g] for g in B128_GROUPS]


def test_fragment_reads_are_bank_conflict_free():
    """ds_read_b128 is served in 4 groups of 16 lanes; a group is conflict-free when its 16 lanes
    touch 16 distinct 16-byte columns of the 256-byte bank row (MI355X_MICROARCH.md, LDS)."""
    for q in range(2):
        for wave in range(4):
            for grp in B128_GROUPS:
                cols_a, cols_b = set(), set()
                for lane in grp:
                    i16, g4 = lane & 15, lane >> 4
                    a_addr = (i16 * 128 + ((g4 ^ swz(i16)) << 4)) ^ (64 * q)
                    pos = wave * 16 + i16
                    b_addr = (pos * 128 + ((g4 ^ swz(pos)) << 4)) ^ (64 * q)
                    cols_a.add((a_addr // 16) % 16)
                    cols_b.add((b_addr // 16) % 16)
                assert len(cols_a) == 16 and len(cols_b) == 16
    # NCHW features: ds_read_b32 in 2 groups of 32 lanes over 32 four-byte banks
    for wave in range(4):
        for half in range(2):
            banks = set()
            for lane in range(32 * half, 32 * half + 32):
                i16, g4 = lane & 15, lane >> 4
                banks.add(((g4 * 1088 + (wave * 16 + i16) * 4) // 4) % 32)
            assert len(banks) == 32
