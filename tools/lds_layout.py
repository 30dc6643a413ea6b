import itertools
GROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],
          [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
          [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59],
          [36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
def cost(rows16, S, F):
    """rows16: LDS row of lane idx 0..15; lane = g*16+idx reads 16 B at row*S*16 + (g^F(row))*16.  returns total LDS cycles (4 = conflict-free)"""
    tot = 0
    for grp in GROUPS:
        banks = {}
        for lane in grp:
            g, idx = lane >> 4, lane & 15
            r = rows16[idx]
            slot = (r * S + (g ^ F(r))) % 16
            banks.setdefault(slot, set()).add((r, g))
        tot += max(len(v) for v in banks.values())
    return tot
def conv1_rows(tile, tap, P):
    rows = []
    for idx in range(16):
        hp = min(tile * 16 + idx, 99)
        hy, hx = divmod(hp, 10)
        rows.append((hy + tap // 3) * P + hx + tap % 3)
    return rows
def conv2_rows(tile, tap, P):
    rows = []
    for idx in range(16):
        op = tile * 16 + idx
        oy, ox = op >> 3, op & 7
        rows.append((oy + tap // 3) * P + ox + tap % 3)
    return rows
Fs = {"none": lambda r: 0, "r>>1&2": lambda r: (r >> 1) & 2, "r>>2&3": lambda r: (r >> 2) & 3, "r>>1&3": lambda r: (r >> 1) & 3,
      "r&3": lambda r: r & 3, "r>>2&1": lambda r: (r >> 2) & 1, "r>>3&3": lambda r: (r >> 3) & 3, "(r>>2&1)*2": lambda r: ((r >> 2) & 1) * 2,
      "(r>>2^r>>3)&3": lambda r: ((r >> 2) ^ (r >> 3)) & 3}
best = []
for S in (4, 5):
    for fn, F in Fs.items():
        w = cost(list(range(16)), S, F)
        for P1 in range(12, 21):
            c1 = sum(cost(conv1_rows(t, tap, P1), S, F) for t in range(7) for tap in range(9)) / 63
            for P2 in range(10, 21):
                c2 = sum(cost(conv2_rows(t, tap, P2), S, F) for t in range(4) for tap in range(9)) / 36
                best.append((c1 + c2 + w, S, fn, P1, P2, w, round(c1, 2), round(c2, 2)))
best.sort()
for b in best[:12]:
    print(b)
print("current:", [b for b in best if b[1] == 5 and b[2] == "none" and b[3] == 12 and b[4] == 10])
