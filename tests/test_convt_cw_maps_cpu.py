"""CPU restatement of the index maps and class tables of the round-5 kernels (csrc/convt_cw.hip; further down csrc/conv3_cw.hip and
csrc/conv4s2d_cw.hip) - no GPU:
  * the patch DMA (wave w brings row block w of every 32-channel chunk; lane -> pixel, 16-byte piece, LDS byte) covers the (4+1) x (16+1)
    window every class reads, with the swizzle the fragment reads undo;
  * every B-fragment ds_read_b128 of every class / tap / tile row lands on the bytes it means and its 16-lane service groups are
    conflict-free;
  * the class / tap / slot tables ARE conv_transpose2d(k3, s2, p1, op1): a numpy evaluation through them equals the direct definition
    (code/ops.py:45-54 of the reference);
  * the wave -> class assignment puts classes 3 + 0 and 1 + 2 on the SIMD pairs, every (class, channel half) exactly once."""
import numpy as np

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]
K_ROW, K_PITCH, K_TH = 64, 24, 4
USED_BLOCKS = ((K_TH + 1) * K_PITCH + 15) // 16
CHUNK_BYTES = USED_BLOCKS * 1024
# Taps<CLS> of the kernel: (slot, dy, dx)
TAPS = {0: [(4, 0, 0)], 1: [(3, 0, 1), (5, 0, 0)], 2: [(1, 1, 0), (7, 0, 0)], 3: [(0, 1, 1), (2, 1, 0), (6, 0, 1), (8, 0, 0)]}


def swz(row, piece):
    return row * K_ROW + ((piece ^ ((row >> 1) & 2)) << 4)


def dma_image(nch):
    """LDS byte -> (chunk, patch row, patch column, logical 16-byte piece) as the eight waves' DMA instructions place them"""
    assert USED_BLOCKS == 8
    lds = {}
    for wid in range(8):
        for u in range(nch):                      # block j = wid + 8 u: row block wid of chunk u
            for lane in range(64):
                drow = wid * 16 + (lane >> 2)
                px, py = drow % K_PITCH, drow // K_PITCH
                valid = py <= K_TH and px <= 16
                piece = (lane & 3) ^ ((drow >> 1) & 2)      # the lane fetches this LOGICAL piece of its pixel's chunk-u channels
                addr = wid * 1024 + u * 8192 + lane * 16    # ... and the DMA puts lane l's 16 bytes at + 16 l
                assert addr not in lds and addr + 16 <= nch * CHUNK_BYTES
                lds[addr] = (u, py, px, piece) if valid else None
    return lds


def test_patch_dma_covers_the_window_and_fragment_reads_hit_it_conflict_free():
    for nch in (2, 4):
        lds = dma_image(nch)
        seen = {v for v in lds.values() if v is not None}
        assert seen == {(c, py, px, pc) for c in range(nch) for py in range(K_TH + 1) for px in range(17) for pc in range(4)}
        for cls, taps in TAPS.items():
            for ci in range(nch):
                for (_, dy, dx) in taps:
                    for b in range(4):            # tile row
                        addrs = []
                        for lane in range(64):
                            idx, g = lane & 15, lane >> 4
                            a = ci * CHUNK_BYTES + swz(b * K_PITCH + idx + dx, g) + dy * K_PITCH * K_ROW
                            assert lds[a] == (ci, b + dy, idx + dx, g), (nch, cls, ci, dy, dx, b, lane)
                            addrs.append(a)
                        for grp in GROUPS:
                            slots = {}
                            for l in grp:
                                slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
                            assert all(len(v) == 1 for v in slots.values()), ("bank conflict", nch, cls, dy, dx, b)


def test_class_tables_are_the_transposed_convolution():
    rng = np.random.default_rng(5)
    cin, cout, H, W = 3, 2, 5, 6
    x = rng.standard_normal((cin, H, W))
    w = rng.standard_normal((cin, cout, 3, 3))      # ConvTranspose2d weight [in, out, kh, kw]
    ref = np.zeros((cout, 2 * H, 2 * W))            # out[2y + ky - 1][2x + kx - 1] += x[y][x] * w[ky][kx]  (stride 2, padding 1, output_padding 1)
    for y in range(H):
        for xx in range(W):
            for ky in range(3):
                for kx in range(3):
                    oy, ox = 2 * y + ky - 1, 2 * xx + kx - 1
                    if 0 <= oy < 2 * H and 0 <= ox < 2 * W:
                        ref[:, oy, ox] += np.einsum("i,io->o", x[:, y, xx], w[:, :, ky, kx])
    # slot s of the forward packing holds kernel tap (ky, kx) = (s // 3, s % 3): the 9-slot table of csrc/convt_mfma.hip / engine.ConvSpec
    got = np.zeros_like(ref)
    xp = np.zeros((cin, H + 1, W + 1))
    xp[:, :H, :W] = x
    for cls, taps in TAPS.items():
        oy, ox = cls >> 1, cls & 1
        for (slot, dy, dx) in taps:
            ky, kx = slot // 3, slot % 3
            for y in range(H):
                for xx in range(W):
                    got[:, 2 * y + oy, 2 * xx + ox] += np.einsum("i,io->o", xp[:, y + dy, xx + dx], w[:, :, ky, kx])
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    assert sorted(s for t in TAPS.values() for (s, _, _) in t) == list(range(9))


def test_wave_roles():
    seen = {}
    for wid in range(8):
        wc = wid & 1
        cls = (2 if (wid >> 2) else 1) if ((wid >> 1) & 1) else (0 if (wid >> 2) else 3)
        seen[(cls, wc)] = wid
    assert len(seen) == 8
    for simd in range(4):                            # waves w and w + 4 share a SIMD
        a = [c for (c, _), w in seen.items() if w in (simd, simd + 4)]
        assert sorted(a) in ([0, 3], [1, 2])         # 5 and 4 taps per SIMD


# ------------------------------------------------------------------------------------------------ csrc/conv3_cw.hip
def test_conv3_cw_patch_dma_and_fragment_reads():
    """csrc/conv3_cw.hip: chunk images padded to 16 one-KiB blocks (block j = wave + 8 u: row block wave + 8 (u & 1) of chunk u >> 1);
    the 10 x 18 patch (origin (ty0 - 1, tx0 - 1)) is covered, every fragment read of wave (wc, rg), tile row b, tap (dy, dx) lands on
    (chunk, row 2 rg + b + dy, column idx + dx, piece g) and is conflict-free; the 32-channel form uses chunk 0 only"""
    TH, PH, PW, CHUNK = 8, 10, 18, 16 * 1024
    for nch in (1, 2):
        lds = {}
        for wid in range(8):
            for e in range(2):
                for cc in range(nch):
                    for lane in range(64):
                        row = (wid + 8 * e) * 16 + (lane >> 2)
                        py, px = row // K_PITCH, row % K_PITCH
                        valid = py < PH and px < PW
                        piece = (lane & 3) ^ (((wid * 16 + (lane >> 2)) >> 1) & 2)   # the kernel takes the swizzle key from row block `wave`
                        assert piece == (lane & 3) ^ ((row >> 1) & 2)              # ... which is the same 128 rows further on
                        addr = cc * CHUNK + (wid + 8 * e) * 1024 + lane * 16
                        assert addr not in lds
                        lds[addr] = (cc, py, px, piece) if valid else None
        seen = {v for v in lds.values() if v is not None}
        assert seen == {(c, py, px, pc) for c in range(nch) for py in range(PH) for px in range(PW) for pc in range(4)}
        for wid in range(8):
            r0 = (wid >> 1) * 2
            for ci in range(nch):
                for so in range(9):
                    for b in range(2):
                        addrs = []
                        for lane in range(64):
                            idx, g = lane & 15, lane >> 4
                            a = ci * CHUNK + swz((r0 + b) * K_PITCH + idx + so % 3, g) + (so // 3) * K_PITCH * K_ROW
                            assert lds[a] == (ci, r0 + b + so // 3, idx + so % 3, g)
                            addrs.append(a)
                        for grp in GROUPS:
                            slots = {}
                            for l in grp:
                                slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
                            assert all(len(v) == 1 for v in slots.values()), ("bank conflict", nch, wid, so, b)


# ------------------------------------------------------------------------------------------------ csrc/conv4s2d_cw.hip
C4D_TAPS = {0: [(5, 1, 1), (7, 1, 0), (13, 0, 1), (15, 0, 0)], 1: [(4, 1, 2), (6, 1, 1), (12, 0, 2), (14, 0, 1)],
            2: [(1, 2, 1), (3, 2, 0), (9, 1, 1), (11, 1, 0)], 3: [(0, 2, 2), (2, 2, 1), (8, 1, 2), (10, 1, 1)]}


def test_conv4s2d_cw_patch_dma_and_fragment_reads():
    """csrc/conv4s2d_cw.hip: 9 one-KiB blocks per chunk (6 x 24 image rows); wave w brings row block w of every chunk, waves 0 .. NCH-1
    block 8 of chunk w; the (4+2) x (16+2) window is covered and every class's fragment reads hit it conflict-free"""
    CHUNK = 9 * 1024
    for nch in (2, 4):
        lds = {}
        for wid in range(8):
            for u in range(nch):
                for lane in range(64):
                    row = wid * 16 + (lane >> 2)
                    py, px = row // K_PITCH, row % K_PITCH
                    addr = u * CHUNK + wid * 1024 + lane * 16
                    assert addr not in lds
                    lds[addr] = (u, py, px, (lane & 3) ^ ((row >> 1) & 2)) if px < 18 else None
            if wid < nch:
                for lane in range(64):
                    row = 8 * 16 + (lane >> 2)
                    py, px = row // K_PITCH, row % K_PITCH
                    addr = wid * CHUNK + 8 * 1024 + lane * 16
                    assert addr not in lds
                    lds[addr] = (wid, py, px, (lane & 3) ^ ((row >> 1) & 2)) if px < 18 else None
        seen = {v for v in lds.values() if v is not None}
        assert seen == {(c, py, px, pc) for c in range(nch) for py in range(6) for px in range(18) for pc in range(4)}
        for cls, taps in C4D_TAPS.items():
            for ci in range(nch):
                for (_, wy, wx) in taps:
                    for b in range(4):
                        addrs = []
                        for lane in range(64):
                            idx, g = lane & 15, lane >> 4
                            a = ci * CHUNK + swz(b * K_PITCH + idx + wx, g) + wy * K_PITCH * K_ROW
                            assert lds[a] == (ci, b + wy, idx + wx, g)
                            addrs.append(a)
                        for grp in GROUPS:
                            slots = {}
                            for l in grp:
                                slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
                            assert all(len(v) == 1 for v in slots.values()), ("bank conflict", nch, cls, wy, wx, b)


def test_conv4s2d_class_tables_are_the_input_gradient_of_the_stride_2_convolution():
    """din[2y+oy][2x+ox] = sum over the class's taps of dout[y + wy - 1][x + wx - 1] * W[ky][kx]: slot s of the 16-slot table is kernel tap
    (ky, kx) = (s // 4, s % 4) (csrc/convt_mfma.hip, PAT 1); checked against the definition of conv2d(k4, s2, p1)'s input-gradient"""
    rng = np.random.default_rng(6)
    cin, cout, OH, OW = 2, 3, 4, 5
    dout = rng.standard_normal((cout, OH, OW))
    w = rng.standard_normal((cout, cin, 4, 4))          # Conv2d weight [out, in, kh, kw]
    ref = np.zeros((cin, 2 * OH, 2 * OW))               # out[y][x] += in[2y + ky - 1][2x + kx - 1] w[ky][kx]  =>  din[2y+ky-1][2x+kx-1] += dout[y][x] w
    for y in range(OH):
        for x in range(OW):
            for ky in range(4):
                for kx in range(4):
                    iy, ix = 2 * y + ky - 1, 2 * x + kx - 1
                    if 0 <= iy < 2 * OH and 0 <= ix < 2 * OW:
                        ref[:, iy, ix] += np.einsum("o,oi->i", dout[:, y, x], w[:, :, ky, kx])
    dp = np.zeros((cout, OH + 2, OW + 2))
    dp[:, 1:-1, 1:-1] = dout
    got = np.zeros_like(ref)
    for cls, taps in C4D_TAPS.items():
        oy, ox = cls >> 1, cls & 1
        for (slot, wy, wx) in taps:
            ky, kx = slot // 4, slot % 4
            for y in range(OH):
                for x in range(OW):
                    got[:, 2 * y + oy, 2 * x + ox] += np.einsum("o,oi->i", dp[:, y + wy, x + wx], w[:, :, ky, kx])
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    assert sorted(s for t in C4D_TAPS.values() for (s, _, _) in t) == list(range(16))
