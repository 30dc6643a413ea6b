"""CPU restatement of csrc/resblock_ws.hip's index maps (patch / W1 DMA -> LDS -> fragment reads): every read lands on the bytes it means,
and every ds_read_b128 lane group is conflict-free."""
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]
def row_to_channel(R):
    tile, r = R >> 4, R & 15
    q, j = r >> 2, r & 3
    return 32 * (tile >> 1) + 8 * q + 4 * (tile & 1) + j
def img_off(row, piece): return row * 64 + ((piece ^ ((row >> 2) & 3)) << 4)
def check(TH):
    kPH = TH + 4; kMain = (kPH * 10 + 15) // 16 * 16; kXB = (kPH + 7) // 8; kPRows = kMain + 16 * kXB; kNPD = kPRows // 8; kHPix = (TH + 2) * 10
    NT1 = (kHPix + 31) // 32; NTW = NT1 // 2
    def patch_row(py, px):
        v = 10 * py + px
        return v if px < 10 else kMain + 16 * (py >> 3) + (v & 15)
    def patch_off(prow, c, piece):
        return (16 * (prow >> 3) + 8 * c + (prow & 7)) * 64 + ((piece ^ ((prow >> 2) & 3)) << 4)
    lds = {}
    issued = [0] * 8
    for wid in range(8):
        NP = (kNPD + 7) // 8 if wid < 4 else (kNPD + 3) // 8
        for k in range(NP):
            j = wid + 8 * k
            assert j < kNPD
            issued[wid] += 1
            lines = set()
            for lane in range(64):
                lrow = lane >> 2; cc = lrow >> 3
                row = 8 * j + (lrow & 7)
                if row < kMain:
                    py = (row * 205) >> 11; px = row - 10 * py; valid = row < kPH * 10
                else:
                    e = row - kMain; e4 = e & 15
                    py = 8 * (e >> 4) + ((5 * (e4 >> 1) + 7) & 7); px = 10 + (e4 & 1); valid = py < kPH
                piece = (lane & 3) ^ ((row >> 2) & 3)
                addr = j * 1024 + lane * 16
                assert addr not in lds
                lds[addr] = (cc, py, px, piece) if valid else None
                if valid:
                    lines.add((py, px))
            assert len(lines) <= 8   # one DMA instruction reads at most 8 pixels = 8 whole 128-byte lines
    assert sum(issued) == kNPD and len(lds) == 2 * kPRows * 4
    seen = {v for v in lds.values() if v is not None}
    assert seen == {(cc, py, px, pc) for cc in range(2) for py in range(kPH) for px in range(12) for pc in range(4)}, TH
    # conv1 B reads
    for pg in range(2):
        for tw in range(NTW):
            for t in range(9):
                for kc in range(2):
                    for s16 in range(2):
                        addrs = []
                        for lane in range(64):
                            l32, hi = lane & 31, lane >> 5
                            n0 = 32 * (pg * NTW + tw) + l32; nn = n0 if n0 < kHPix else n0 - 32
                            hy = (nn * 205) >> 11; hx = nn - 10 * hy
                            assert hy == nn // 10
                            a = kc * 512 + (patch_off(patch_row(hy + t // 3, hx + t % 3), 0, hi) ^ (32 * s16))
                            assert lds[a] == (kc, hy + t // 3, hx + t % 3, 2 * s16 + hi), (TH, lane, t)
                            addrs.append(a)
                        for g in GROUPS:
                            slots = {}
                            for l in g:
                                slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
                            assert all(len(v) == 1 for v in slots.values()), ("bank conflict", TH, pg, tw, t, g)
    # h image (pitch 24): conv1's finalise writes pieces 2 hi, 2 hi + 1 of chunk rt; conv2's lane l reads output pixel (l / 8, l % 8)
    # of its 32-pixel tile under each tap, piece 2 s16 + hi of chunk kc; the skip read takes piece 2 hi + kc of patch pixel (oy + 2, ox + 2)
    kHP = 24; kHChunk = (TH + 2) * kHP * 64
    h = {}
    for rt in range(2):
        for n0 in range(kHPix):
            hy, hx = n0 // 10, n0 % 10
            for hi in range(2):
                for pc in (2 * hi, 2 * hi + 1):
                    a = rt * kHChunk + img_off(hy * kHP + hx, pc)
                    assert a not in h and a + 16 <= 2 * kHChunk
                    h[a] = (rt, hy, hx, pc)
    for kc in range(2):
        for t in range(TH * 8 // 32):
            for tt in range(9):
                for s16 in range(2):
                    addrs = []
                    for lane in range(64):
                        l32, hi = lane & 31, lane >> 5
                        oy, ox = 4 * t + (l32 >> 3), l32 & 7
                        a = kc * kHChunk + (img_off((oy + tt // 3) * kHP + ox + tt % 3, hi) ^ (32 * s16))
                        assert h[a] == (kc, oy + tt // 3, ox + tt % 3, 2 * s16 + hi)
                        addrs.append(a)
                        for rt in range(2):
                            assert lds[patch_off(10 * (oy + 2) + ox + 2, rt, 2 * hi + kc)] == (rt, oy + 2, ox + 2, 2 * hi + kc)
                    for g in GROUPS:
                        assert len({(addrs[l] // 16) % 16 for l in g}) == 16, ("bank conflict (h)", TH, t, tt, g)
    # W1
    w = {}
    for wid in range(8):
        u, cw = wid & 3, wid >> 2
        for lane in range(64):
            mp = 16 * u + (lane >> 2)
            rt, m = mp >> 5, mp & 31
            j, hm, e = m >> 3, (m >> 2) & 1, m & 3
            R = 16 * (2 * rt + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e
            piece = (lane & 3) ^ ((mp >> 2) & 3)
            for t in range(9):
                addr = (2 * t + cw) * 4096 + u * 1024 + lane * 16
                assert addr not in w
                w[addr] = (t, cw, R, piece)
    assert len(w) == 18 * 256
    for rt in range(2):
        for t in range(9):
            for kc in range(2):
                for s16 in range(2):
                    addrs = []
                    for lane in range(64):
                        l32, hi = lane & 31, lane >> 5
                        a = (2 * t + kc) * 4096 + (img_off(32 * rt + l32, hi) ^ (32 * s16))
                        tt, cc, R, piece = w[a]
                        assert (tt, cc, piece) == (t, kc, 2 * s16 + hi)
                        # D row m = l32 of this lane's A row must be channel 32 rt + 16 hm + 4 j + e, so that accumulator register
                        # i = 4 j + e of lane (col, hi) [row 8 j + 4 hi + e] is channel 32 rt + 16 hi + i
                        j, hm, e = l32 >> 3, (l32 >> 2) & 1, l32 & 3
                        assert row_to_channel(R) == 32 * rt + 16 * hm + 4 * j + e
                        addrs.append(a)
                    for g in GROUPS:
                        assert len({(addrs[l] // 16) % 16 for l in g}) == 16
    # conv2's A-fragments: two whole-row loads per tap (X, Y), v_permlane16_swap (odd 16-lane rows of X <-> even rows of Y)
    for rt in range(2):
        X, Y = {}, {}
        for lane in range(64):
            l32, hi = lane & 31, lane >> 5
            m = l32 & 15; j, hm, e = m >> 3, (m >> 2) & 1, m & 3
            R = 16 * (2 * rt + (j & 1)) + 4 * (2 * hm + (j >> 1)) + e
            pc = 2 * (l32 >> 4) + hi
            X[lane] = (R, pc); Y[lane] = (R + 4, pc)
        assert len({(r >> 1) for r, _ in X.values()}) == 8 and len({(r >> 1) for r, _ in Y.values()}) == 8   # 8 lines each
        F0, F1 = dict(X), dict(Y)
        for lane in range(64):
            if (lane >> 4) & 1:          # odd row of X takes the even row below it of Y, and gives its own to it
                F0[lane], F1[lane - 16] = Y[lane - 16], X[lane]
        for lane in range(64):
            l32, hi = lane & 31, lane >> 5
            j, hm, e = l32 >> 3, (l32 >> 2) & 1, l32 & 3
            for s16, F in ((0, F0), (1, F1)):
                R, pc = F[lane]
                assert pc == 2 * s16 + hi and row_to_channel(R) == 32 * rt + 16 * hm + 4 * j + e, (lane, s16, F[lane])
    for i in range(16):
        for hi in range(2):
            m = (i & 3) + 8 * (i >> 2) + 4 * hi
            j, hm, e = m >> 3, (m >> 2) & 1, m & 3
            assert 16 * hm + 4 * j + e == 16 * hi + i
    return kPRows


def test_resblock_ws_lds_maps():
    """csrc/resblock_ws.hip: the patch image has no wasted row for 8 x 4 tiles (96 rows = 96 pixels), 160 for 8 x 8"""
    assert check(4) == 96 and check(8) == 160


def patch_off(prow, c, piece):
    return (16 * (prow >> 3) + 8 * c + (prow & 7)) * 64 + ((piece ^ ((prow >> 2) & 3)) << 4)


def conflict_free(addrs):
    for g in GROUPS:
        slots = {}
        for l in g:
            slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
        if not all(len(v) == 1 for v in slots.values()):
            return False
    return True


def check_pair():
    """csrc/exp/resblock2_ws.hip (two blocks per launch, 8 x 4 tiles): patch (16 x 12, read by the 14-wide h1 region), h1 (pitch 28, read by
    the 12-wide a1 region), a1 (12 x 8 image written by conv2A's epilogue, read by the 10-wide h2 region), h2 (pitch 24)"""
    TH = 4
    div = {8: lambda n: n >> 3, 10: lambda n: (n * 205) >> 11, 12: lambda n: (n * 171) >> 11, 14: lambda n: (n * 4682) >> 16}
    for wr, top in ((10, 96), (12, 144), (14, 224)):
        assert all(div[wr](n) == n // wr for n in range(top))
    def src_row(wr, py, px, main):
        v = wr * py + px
        return v if px < wr else main + 16 * (py >> 3) + (v & 15)
    # ---- patch DMA (reader WR = 14)
    kP0H = TH + 8; kP0Main = (14 * kP0H + 15) // 16 * 16; kP0Rows = 224; kNPD = kP0Rows // 8
    assert kP0Main == 176 and kNPD == 28
    lds = {}
    for wid in range(8):
        NP = (kNPD + 7) // 8 if wid < 4 else (kNPD + 3) // 8
        for k in range(NP):
            j = wid + 8 * k
            assert j < kNPD
            for lane in range(64):
                lrow = lane >> 2; cc = lrow >> 3
                row = 8 * j + (lrow & 7)
                if row < kP0Main:
                    py = div[14](row); px = row - 14 * py; valid = row < 14 * kP0H
                else:
                    e = row - kP0Main; e4 = e & 15
                    py = 8 * (e >> 4) + ((7 * (e4 >> 1) + 7) & 7); px = 14 + (e4 & 1); valid = py < kP0H
                piece = (lane & 3) ^ ((row >> 2) & 3)
                addr = j * 1024 + lane * 16
                assert addr not in lds
                lds[addr] = (cc, py, px, piece) if valid else None
    assert len(lds) == kP0Rows * 8
    assert {v for v in lds.values() if v} == {(cc, py, px, pc) for cc in range(2) for py in range(kP0H) for px in range(16) for pc in range(4)}
    for (py, px) in [(py, px) for py in range(kP0H) for px in range(16)]:
        for cc in range(2):
            for pc in range(4):
                assert lds[patch_off(src_row(14, py, px, kP0Main), cc, pc)] == (cc, py, px, pc)
    # ---- conv1 of block A: region 14 x 10 = 140 pixels, tiles 0-2 / 3-4
    kH1Pix = 140
    for t0, ntw in ((0, 3), (3, 2)):
        for tw in range(ntw):
            for t in range(9):
                for kc in range(2):
                    for s16 in range(2):
                        addrs = []
                        for lane in range(64):
                            l32, hi = lane & 31, lane >> 5
                            n0 = 32 * (t0 + tw) + l32; nn = n0 if n0 < kH1Pix else n0 - 32
                            hy = div[14](nn); hx = nn - 14 * hy
                            a = kc * 512 + (patch_off(src_row(14, hy + t // 3, hx + t % 3, kP0Main), 0, hi) ^ (32 * s16))
                            assert lds[a] == (kc, hy + t // 3, hx + t % 3, 2 * s16 + hi)
                            addrs.append(a)
                        assert conflict_free(addrs), ("conv1A", t0, tw, t)
    # ---- h1 (pitch 28): written at img_off(28 hy + hx, pieces 2 hi, 2 hi + 1) of chunk rt; conv2A lanes = a1 region pixels (12 wide)
    kH1Chunk = 28 * 10 * 64
    h1 = {}
    for rt in range(2):
        for n0 in range(kH1Pix):
            hy, hx = n0 // 14, n0 % 14
            for pc in range(4):
                a = rt * kH1Chunk + img_off(hy * 28 + hx, pc)
                assert a not in h1 and a + 16 <= 2 * kH1Chunk
                h1[a] = (rt, hy, hx, pc)
    a1 = {}
    kA1Main = 80
    for kc in range(2):
        for t in range(3):
            for tt in range(9):
                for s16 in range(2):
                    addrs = []
                    for lane in range(64):
                        l32, hi = lane & 31, lane >> 5
                        nn = 32 * t + l32; oy = div[12](nn); ox = nn - 12 * oy
                        assert oy < 8
                        a = kc * kH1Chunk + (img_off((oy + tt // 3) * 28 + ox + tt % 3, hi) ^ (32 * s16))
                        assert h1[a] == (kc, oy + tt // 3, ox + tt % 3, 2 * s16 + hi)
                        addrs.append(a)
                        for rt in range(2):   # skip read from the patch, a1 write
                            assert lds[patch_off(14 * (oy + 2) + ox + 2, rt, 2 * hi + kc)] == (rt, oy + 2, ox + 2, 2 * hi + kc)
                            a1[patch_off(src_row(10, oy, ox, kA1Main), rt, 2 * hi + kc)] = (rt, oy, ox, 2 * hi + kc)
                    assert conflict_free(addrs), ("conv2A", t, tt)
    assert len(a1) == 96 * 8 and max(a1) + 16 <= 96 * 128
    # ---- conv1 of block B: region 10 x 6 from the a1 image, one tile per wave
    for pg in range(2):
        for t in range(9):
            for kc in range(2):
                for s16 in range(2):
                    addrs = []
                    for lane in range(64):
                        l32, hi = lane & 31, lane >> 5
                        n0 = 32 * pg + l32; nn = n0 if n0 < 60 else n0 - 32
                        hy = div[10](nn); hx = nn - 10 * hy
                        a = kc * 512 + (patch_off(src_row(10, hy + t // 3, hx + t % 3, kA1Main), 0, hi) ^ (32 * s16))
                        assert a1[a] == (kc, hy + t // 3, hx + t % 3, 2 * s16 + hi)
                        addrs.append(a)
                    assert conflict_free(addrs), ("conv1B", pg, t)
    # skip read of block B: a1 pixel (oy + 2, ox + 2)
    for lane in range(64):
        l32, hi = lane & 31, lane >> 5
        oy, ox = l32 >> 3, l32 & 7
        for rt in range(2):
            for kc in range(2):
                assert a1[patch_off(10 * (oy + 2) + ox + 2, rt, 2 * hi + kc)] == (rt, oy + 2, ox + 2, 2 * hi + kc)
    # LDS budget
    kP0 = 18 * 4096; kH1 = kP0 + kP0Rows * 128; kA1 = kH1 + 2 * kH1Chunk
    assert kA1 + 96 * 128 == 150528 <= 160 * 1024
    for base in (kP0, kH1, kA1, kH1 + kH1Chunk, kH1 + 24 * 6 * 64):
        assert base & 32 == 0   # (fragment offsets toggle the chunk half with XOR 32)
    return True


def test_resblock2_ws_lds_maps():
    assert check_pair()
