"""CPU restatement of the index maps of csrc/conv_s2_cw.hip (round 6; no GPU):
  * the phase DMA (wave w brings 1-KiB blocks w, w + 8, w + 16 of both chunks, KS = 4: waves 0 / 1 block 24 of chunk 0 / 1; lane -> input
    pixel with the columns de-interleaved by parity, 16-byte piece, LDS byte) covers the (2*4 + KS - 2) x (2*16 + KS - 2) window;
  * every B-fragment ds_read_b128 (tile row b, tap ky kx, chunk) lands on input pixel (2b + ky, 2 idx + kx) of the window, logical piece
    g, and its 16-lane service groups are conflict-free;
  * the four wave groups' k-step ranges partition the (tap, chunk) steps, SIMD pairs balanced;
  * the exchange slots: every (channel half, row, source != row) has its own 2 KiB; writer and reader agree;
  * the tables ARE the convolutions: a numpy evaluation through (step -> tap -> slot, patch maps) equals conv2d(k4, s2, p1) /
    the conv-transpose's input-gradient (code/models.py:90-94, code/ops.py:45-54 of the reference)."""
import numpy as np

GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]
K_ROW, K_PITCH, K_ODD, K_TH = 64, 40, 20, 4


def geo(ks):
    pr = 2 * K_TH + ks - 2
    rows = pr * K_PITCH
    kb = (rows + 15) // 16
    first = [0, 8, 16, 24, 32] if ks == 4 else [0, 5, 9, 13, 18]
    return pr, kb, kb * 1024, first


def swz(row, piece):
    return row * K_ROW + ((piece ^ ((row >> 1) & 2)) << 4)


def dma_image(ks):
    """LDS byte of a phase buffer -> (chunk, window row, window column, logical piece) or None (pitch padding: the zero page)"""
    pr, kb, chunk_bytes, _ = geo(ks)
    lds = {}
    for wid in range(8):
        blocks = [(wid + 8 * e, (0, 1)) for e in range(3) if wid + 8 * e < kb]
        if ks == 4 and wid < 2:
            blocks.append((24, (wid,)))
        for rb, chunks in blocks:
            for lane in range(64):
                row = rb * 16 + (lane >> 2)
                py, c = divmod(row, K_PITCH)
                par = 1 if c >= K_ODD else 0
                jj = c - K_ODD * par
                valid = rb < kb and py < pr and jj < 17
                piece = (lane & 3) ^ ((lane >> 3) & 2)
                assert ((lane >> 3) & 2) == ((row >> 1) & 2)          # the kernel's lane-only form of the swizzle key
                for u in chunks:
                    addr = u * chunk_bytes + rb * 1024 + lane * 16
                    assert addr not in lds and addr + 16 <= 2 * chunk_bytes
                    lds[addr] = (u, py, 2 * jj + par, piece) if valid else None
    return lds


def test_phase_dma_covers_the_window_and_fragment_reads_hit_it_conflict_free():
    for ks in (3, 4):
        pr, kb, chunk_bytes, _ = geo(ks)
        lds = dma_image(ks)
        assert len(lds) == 2 * kb * 64                                 # every block of both chunks exactly once
        need = {(u, py, px, pc) for u in range(2) for py in range(pr) for px in range(2 * 16 + ks - 2) for pc in range(4)}
        assert need <= {v for v in lds.values() if v is not None}
        for cc in range(2):
            for ky in range(ks):
                for kx in range(ks):
                    for b in range(4):
                        addrs = []
                        for lane in range(64):
                            idx, g = lane & 15, lane >> 4
                            a = cc * chunk_bytes + swz(K_ODD * (kx & 1) + (kx >> 1) + idx, g) + (2 * b + ky) * K_PITCH * K_ROW
                            assert lds[a] == (cc, 2 * b + ky, 2 * idx + kx, g), (ks, cc, ky, kx, b, lane)
                            addrs.append(a)
                        for grp in GROUPS:
                            slots = {}
                            for l in grp:
                                slots.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
                            assert max(len(v) for v in slots.values()) == 1


def test_k_step_ranges_partition_the_reduction_and_balance_the_simds():
    for ks in (3, 4):
        first = geo(ks)[3]
        ns = 2 * ks * ks
        steps = [s for kg in range(4) for s in range(first[kg], first[kg + 1])]
        assert steps == list(range(ns))
        cnt = [first[kg + 1] - first[kg] for kg in range(4)]
        assert cnt[0] + cnt[2] == cnt[1] + cnt[3]                      # wave groups kg and kg + 2 share a SIMD (waves w, w + 4)
        assert max(cnt) <= (8 if ks == 4 else 5)


def test_exchange_slots():
    """both forms: NWC = 2 channel halves x NKG = 4 reduction groups (a wave finishes tile row kg) and NWC = 4 quarters x NKG = 2 halves
    (rows kg, kg + 2): every (channel group, row, source != owner) has its own 2 KiB inside the 48-KB image; writer and reader agree"""
    for nwc in (2, 4):
        nkg = 8 // nwc
        used = {}
        for wc in range(nwc):
            for kg in range(nkg):                  # writer
                for b in range(4):
                    if b % nkg == kg:
                        continue
                    owner = b % nkg
                    slot = kg if kg < owner else kg - 1
                    for a in range(2):
                        off = (((wc * 4 + b) * (nkg - 1) + slot) * 2 + a) * 1024
                        assert off not in used and off + 1024 <= 2 * 4 * 3 * 2048
                        used[off] = (wc, b, kg, a)
        for wc in range(nwc):
            for kg in range(nkg):                  # reader: rows kg, kg + nkg, ...; source k
                for row in range(kg, 4, nkg):
                    for k in range(nkg):
                        if k == kg:
                            continue
                        slot = k if k < kg else k - 1
                        for a in range(2):
                            assert used[(((wc * 4 + row) * (nkg - 1) + slot) * 2 + a) * 1024] == (wc, row, k, a)
        assert len(used) == nwc * 4 * (nkg - 1) * 2


def test_wide_form_is_chosen_where_a_reduction_half_fits_the_registers():
    """host rule of go_s2cw: 128 output channels per workgroup iff Cout % 128 == 0 and KS^2 * Cin / 32 <= 36 k-steps (a half = 18 steps
    x 8 VGPRs = 144): every layer of the step but the 128 -> 128 4x4 one; the halves of the k-steps partition the reduction"""
    wide = lambda ks, cin, cout: cout % 128 == 0 and ks * ks * (cin // 32) <= 36
    assert wide(3, 128, 128) and wide(3, 64, 128) and wide(4, 64, 128)
    assert not wide(4, 128, 128) and not wide(4, 64, 64) and not wide(3, 128, 64)
    for ks in (3, 4):
        ns = 2 * ks * ks
        assert ns % 2 == 0 and [(ns // 2) * kg for kg in range(3)] == [0, ns // 2, ns]


def _through_the_tables(x, wslots, ks, oh, ow):
    """out[y][x][co] through the kernel's decomposition: steps -> (tap, chunk) -> window pixel (2b + ky, 2 idx + kx); x [IH][IW][64]"""
    ih, iw, _ = x.shape
    first = geo(ks)[3]
    out = np.zeros((oh, ow, wslots.shape[1]))
    for ty0 in range(0, oh, K_TH):
        for tx0 in range(0, ow, 16):
            part = np.zeros((4, K_TH, 16, wslots.shape[1]))
            for kg in range(4):
                for st in range(first[kg], first[kg + 1]):
                    tap, cc = st >> 1, st & 1
                    ky, kx = divmod(tap, ks)
                    for b in range(K_TH):
                        for idx in range(16):
                            iy, ix = 2 * ty0 - 1 + 2 * b + ky, 2 * tx0 - 1 + 2 * idx + kx
                            if 0 <= iy < ih and 0 <= ix < iw:
                                part[kg, b, idx] += wslots[tap][:, 32 * cc:32 * cc + 32] @ x[iy, ix, 32 * cc:32 * cc + 32]
            tot = ((part[0] + part[1]) + part[2]) + part[3]
            for b in range(K_TH):
                for idx in range(16):
                    if ty0 + b < oh and tx0 + idx < ow:
                        out[ty0 + b, tx0 + idx] = tot[b, idx]
    return out


def test_the_tables_are_the_two_convolutions():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    # 4x4 stride-2 padding-1 forward: slot ky * 4 + kx holds W[:, :, ky, kx]
    x, w = rng.standard_normal((1, 64, 10, 36)), rng.standard_normal((8, 64, 4, 4))
    ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), None, 2, 1)[0].permute(1, 2, 0).numpy()
    got = _through_the_tables(x[0].transpose(1, 2, 0), np.stack([w[:, :, t // 4, t % 4] for t in range(16)]), 4, 5, 18)
    np.testing.assert_allclose(got, ref, rtol=1e-9, atol=1e-9)
    # conv-transpose k3 s2 p1 op1, input-gradient: din[y][x][ci] = sum_{ky,kx} W[ci, co, ky, kx] . dout[2y + ky - 1][2x + kx - 1][co]
    xin = torch.from_numpy(rng.standard_normal((1, 8, 6, 17))).requires_grad_(True)
    wt = torch.from_numpy(rng.standard_normal((8, 64, 3, 3)))
    dout = rng.standard_normal((1, 64, 12, 34))
    F.conv_transpose2d(xin, wt, None, 2, 1, 1).backward(torch.from_numpy(dout))
    got = _through_the_tables(dout[0].transpose(1, 2, 0), np.stack([wt.numpy()[:, :, t // 3, t % 3] for t in range(9)]), 3, 6, 17)
    np.testing.assert_allclose(got, xin.grad[0].permute(1, 2, 0).numpy(), rtol=1e-9, atol=1e-9)
