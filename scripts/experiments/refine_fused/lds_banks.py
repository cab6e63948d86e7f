#!/usr/bin/env python
"""Bank-conflict search for the fused refinement candidate's conv3 operand reads (ds_read_b128).  Model (MI355X guide + this repo's
earlier brute-force checks): a wave64 b128 read is served in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the
same +32; a group costs as many passes as the maximum number of DISTINCT 16-byte slots that fall on the same 16-byte column of the
256-byte bank row (same address = broadcast = free).

Operand: x16 split planes in LDS, pixel (row, col) of a 20 x 20 patch, 16 channels = two 16-byte blocks (cb = 0, 1) per plane.
Lane (i = lane & 15, kb = lane >> 4) of M-tile t reads block q = 4 ks + kb -> tap = q >> 1, cb = q & 1, pixel m = 16 t + i of the
18 x 18 conv3 patch (linear), i.e. patch position (m // 18 + dy, m % 18 + dx).  Layouts tried: byte address =
(row * ROWP + col) * PIXP + cb * CBOFF, all multiples of 16."""
import itertools

GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def passes(addrs):
    tot = 0
    for g in GROUPS:
        cols = {}
        for l in g:
            a = addrs[l]
            cols.setdefault((a // 16) % 16, set()).add(a)
        tot += max(len(v) for v in cols.values())
    return tot  # 4 = conflict free


def cost(ROWP, PIXP, CBOFF):
    total, n = 0, 0
    for t in range(21):
        for ks in range(5):
            addrs = []
            for lane in range(64):
                i, kb = lane & 15, lane >> 4
                q = min(4 * ks + kb, 17)
                tap, cb = q >> 1, q & 1
                dy, dx = divmod(tap, 3)
                m = min(16 * t + i, 323)
                r, c = divmod(m, 18)
                addrs.append(((r + dy) * ROWP + (c + dx)) * PIXP + cb * CBOFF)
            total += passes(addrs)
            n += 1
    return total / (4.0 * n)


best = []
for ROWP, PIXP in itertools.product(range(20, 29), (32, 48, 64)):
    for CBOFF in (16, 32):
        if CBOFF >= PIXP:
            continue
        best.append((cost(ROWP, PIXP, CBOFF), ROWP, PIXP, CBOFF))
# planar per cb: address = cb * PLANE + (row * ROWP + col) * 16
for ROWP in range(20, 29):
    for PAD in (0, 16, 32, 48, 64, 80, 96, 112):
        PLANE = 20 * ROWP * 16 + PAD
        tot, n = 0, 0
        for t in range(21):
            for ks in range(5):
                addrs = []
                for lane in range(64):
                    i, kb = lane & 15, lane >> 4
                    q = min(4 * ks + kb, 17)
                    tap, cb = q >> 1, q & 1
                    dy, dx = divmod(tap, 3)
                    m = min(16 * t + i, 323)
                    r, c = divmod(m, 18)
                    addrs.append(cb * PLANE + ((r + dy) * ROWP + (c + dx)) * 16)
                tot += passes(addrs)
                n += 1
        best.append((tot / (4.0 * n), ROWP, "planar", PAD))
best.sort(key=lambda b: b[0])
for b in best[:12]:
    print("relative LDS cycles %.3f  ROWP %s  PIXP/planar %s  CBOFF/pad %s" % b)
