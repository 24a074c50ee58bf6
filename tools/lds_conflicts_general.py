"""Bank-conflict model of the general kernel's LDS access pattern (thread g of a track reads/writes sequence
idx(g, q, phase) of 8-byte SoA arrays) and search for an additive index skew  s = idx + (idx >> a) [+ (idx >> b)]
that minimises conflicts for every (S, F).  ds_read_b64: 32-lane groups, 8-byte unit mod 32; ds_write_b64: 16-lane
groups, unit mod 16 (MI355X_MICROARCH.md LDS table)."""
import itertools
import sys


def tables(S, NS, F):
    pw = [S ** i for i in range(F + 1)]
    G, E = S ** NS, S ** F
    NG = E // G
    P = F // __import__("math").gcd(F, NS)
    base = [[0] * NG for _ in range(P)]
    off = [[0] * G for _ in range(P)]
    for ph in range(P):
        h = (1 + ph * NS) % F
        for q in range(G):
            r, o = q, 0
            for j in range(NS):
                o += (r % S) * pw[(h + j) % F]
                r //= S
            off[ph][q] = o
        for g in range(NG):
            r, b = g, 0
            for i in range(F - NS):
                b += (r % S) * pw[(h + NS + i) % F]
                r //= S
            base[ph][g] = b
    return base, off, NG, G, P


def cost(S, NS, F, f):
    """average LDS cycles per access relative to conflict-free (1.0), reads and writes"""
    base, off, NG, G, P = tables(S, NS, F)
    rd = wr = n = 0
    for ph in range(P):
        for q in range(G):
            for g0 in range(0, NG, 32):
                lanes = [f(base[ph][g] + off[ph][q]) for g in range(g0, min(g0 + 32, NG))]
                cnt = {}
                for a in lanes:
                    cnt[a % 32] = cnt.get(a % 32, set()) | {a}
                rd += max(len(v) for v in cnt.values())
                w = 0
                for k in range(0, len(lanes), 16):
                    cnt = {}
                    for a in lanes[k:k + 16]:
                        cnt[a % 16] = cnt.get(a % 16, set()) | {a}
                    w += max(len(v) for v in cnt.values())
                wr += w / max(1, (len(lanes) + 15) // 16)
                n += 1
    return rd / n, wr / n


if __name__ == "__main__":
    cands = {"identity": lambda i: i}
    for a in (3, 4, 5, 6):
        cands["i+(i>>%d)" % a] = (lambda a: lambda i: i + (i >> a))(a)
        for b in (6, 7, 8, 9, 10):
            if b > a:
                cands["i+(i>>%d)+(i>>%d)" % (a, b)] = (lambda a, b: lambda i: i + (i >> a) + (i >> b))(a, b)
    configs = [(4, 1, 3), (4, 1, 4), (4, 1, 5), (4, 1, 6), (3, 1, 4), (3, 1, 5), (3, 1, 6), (3, 1, 7), (2, 1, 6), (2, 1, 8), (2, 1, 10), (5, 1, 4), (6, 1, 4)]
    res = {k: [] for k in cands}
    for cfg in configs:
        for k, f in cands.items():
            res[k].append(cost(*cfg, f))
    print("config".ljust(22), " ".join("S%dF%d" % (c[0], c[2]) for c in configs))
    for k, v in sorted(res.items(), key=lambda kv: sum(r + w for r, w in kv[1])):
        print(k.ljust(22), " ".join("%.1f/%.1f" % rw for rw in v))


def cost_entry(S, NS, F, f):
    """entry-parallel kernel: lane = (group g, member q), q fastest, GP = G rounded up to a power of two."""
    base, off, NG, G, P = tables(S, NS, F)
    GP = 4
    while GP < G:
        GP <<= 1
    rd = wr = n = 0
    for ph in range(P):
        lanes_all = []
        for g in range(NG):
            for q in range(GP):
                lanes_all.append(f(base[ph][g] + off[ph][q]) if q < G else None)
        for k0 in range(0, len(lanes_all), 32):
            lanes = [a for a in lanes_all[k0:k0 + 32] if a is not None]
            if not lanes:
                continue
            cnt = {}
            for a in lanes:
                cnt[a % 32] = cnt.get(a % 32, set()) | {a}
            rd += max(len(v) for v in cnt.values())
            w = 0
            chunks = [[a for a in lanes_all[k:k + 16] if a is not None] for k in range(k0, min(k0 + 32, len(lanes_all)), 16)]
            for ch in chunks:
                cnt = {}
                for a in ch:
                    cnt[a % 16] = cnt.get(a % 16, set()) | {a}
                w += max([len(v) for v in cnt.values()] or [0])
            wr += w / max(1, len(chunks))
            n += 1
    return rd / n, wr / n


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "entry":
    sk = lambda i: i + (i >> 5) + (i >> 10)
    for cfg in [(4, 3, 4), (4, 3, 5), (4, 2, 3), (4, 2, 4), (2, 2, 3), (2, 2, 6), (2, 3, 4), (2, 3, 7), (3, 2, 3), (3, 3, 4)]:
        print(cfg, "identity %.1f/%.1f" % cost_entry(*cfg, lambda i: i), "skew %.1f/%.1f" % cost_entry(*cfg, sk))
