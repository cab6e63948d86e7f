#!/usr/bin/env python
"""Bank-conflict search for the fused stem + conv2 candidate (same lane-group model as ../refine_fused/lds_banks.py).
P3: conv1's B operand = 16 linear pixels of the 19 x 35 patch, read from the conv0 planes [21 rows][RP0 px][16 B] at tap (dy, dx),
    block q = 4 ks + kb (clamped to 8).
P4: conv2's operand = output row 2 w + t, 16 outputs at stride 2, tap q = 4 ks + kb (clamped to 24) of the 5 x 5 window, read from the
    conv1 planes [19 rows][RP1 px][16 B]."""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def passes(addrs):
    tot = 0
    for g in GROUPS:
        cols = {}
        for l in g:
            cols.setdefault((addrs[l] // 16) % 16, set()).add(addrs[l])
        tot += max(len(v) for v in cols.values())
    return tot


def p3(RP0):
    tot, n = 0, 0
    for t in range(42):
        for ks in range(3):
            a = []
            for lane in range(64):
                i, kb = lane & 15, lane >> 4
                q = min(4 * ks + kb, 8)
                dy, dx = divmod(q, 3)
                m = min(16 * t + i, 664)
                r, c = divmod(m, 35)
                a.append(((r + dy) * RP0 + c + dx) * 16)
            tot += passes(a)
            n += 1
    return tot / (4.0 * n)


def p4(RP1):
    tot, n = 0, 0
    for row in range(8):
        for ks in range(7):
            a = []
            for lane in range(64):
                i, kb = lane & 15, lane >> 4
                q = min(4 * ks + kb, 24)
                dy, dx = divmod(q, 5)
                a.append(((2 * row + dy) * RP1 + 2 * i + dx) * 16)
            tot += passes(a)
            n += 1
    return tot / (4.0 * n)


print("P3 (conv1 operand reads), relative LDS cycles by conv0-plane row pitch:", {rp: round(p3(rp), 3) for rp in range(37, 49)})
print("P4 (conv2 operand reads), relative LDS cycles by conv1-plane row pitch:", {rp: round(p4(rp), 3) for rp in range(35, 49)})
