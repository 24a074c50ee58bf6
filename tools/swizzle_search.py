"""Search a GF(2)-linear storage swizzle s = B.w (7x7 bit matrix) for the two-state fast path so that, in every
phase h and for both members q, (a) each 32-lane group of a wave touches 32 distinct 8-byte bank pairs
(ds_read_b64 / b32 banking: index mod 32) and (b) each 16-lane group touches 16 distinct 8-byte units mod 16
(ds_write_b64 banking).  w = ts * 2^F + idx is the wave-level sequence index, idx = rot_F(g, h+1) | q << h."""
import itertools
import random
import sys


def rank_gf2(rows):
    rows = list(rows)
    r = 0
    for bit in range(8):
        piv = None
        for i in range(r, len(rows)):
            if (rows[i] >> bit) & 1:
                piv = i
                break
        if piv is None:
            continue
        rows[r], rows[piv] = rows[piv], rows[r]
        for i in range(len(rows)):
            if i != r and (rows[i] >> bit) & 1:
                rows[i] ^= rows[r]
        r += 1
    return r


def apply(cols, w):
    s = 0
    j = 0
    while w:
        if w & 1:
            s ^= cols[j]
        w >>= 1
        j += 1
    return s


def lanes_to_w(F, h, q, lane):
    E, NG = 1 << F, 1 << (F - 1)
    ts, g = lane // NG, lane % NG
    base = ((g << (h + 1)) | (g >> (F - 1 - h))) & (E - 1)
    return ts * E + (base | (q << h))


def score(cols, F):
    bad = 0
    for h in range(F):
        for q in (0, 1):
            s = [apply(cols, lanes_to_w(F, h, q, l)) for l in range(64)]
            for g0 in (0, 32):
                bad += 32 - len(set(x % 32 for x in s[g0:g0 + 32]))
            for g0 in range(0, 64, 16):
                bad += 16 - len(set(x % 16 for x in s[g0:g0 + 16]))
    return bad


def search(F, iters=200000, seed=0):
    rng = random.Random(seed)
    best = None
    cols = [1 << j for j in range(7)]
    cur = score(cols, F)
    best = (cur, cols[:])
    for it in range(iters):
        c2 = cols[:]
        j = rng.randrange(7)
        c2[j] ^= 1 << rng.randrange(7)
        if rank_gf2(c2) < 7:
            continue
        sc = score(c2, F)
        if sc <= cur or rng.random() < 0.02:
            cols, cur = c2, sc
            if cur < best[0]:
                best = (cur, cols[:])
                if cur == 0:
                    break
    return best


if __name__ == "__main__":
    for F in (4, 5, 6, 7):
        ident = score([1 << j for j in range(7)], F)
        xor31 = None
        b = search(F)
        print("F=%d identity conflicts=%d best=%d cols=%s" % (F, ident, b[0], [hex(c) for c in b[1]]))
