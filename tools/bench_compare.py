#!/usr/bin/env python3
"""Compares the timing fields of two bench.py JSON lines (e.g. profiles/r04_bench_final.json against a fresh run) and lists what moved by
more than a tolerance - the round-4 regression of the threshold-fusion apply kernel (2.65 -> 3.8 ms) sat unnoticed behind the headline for
half a round.   usage: tools/bench_compare.py OLD.json NEW.json [tolerance, default 0.06]"""
import json
import sys


def load(p):
    return json.loads(open(p).read().strip().split("\n")[-1])


def walk(d, pre=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from walk(v, pre + k + ".")
        elif isinstance(v, (int, float)) and not isinstance(v, bool) and any(s in k for s in ("ms", "seconds", "per_s", "value")) and "bytes" not in k:
            yield pre + k, float(v)


old, new = load(sys.argv[1]), load(sys.argv[2])
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.06
a, b = dict(walk(old)), dict(walk(new))
bad = 0
for k in sorted(set(a) & set(b)):
    if a[k] <= 0 or "cpu_baseline" in k or "speedup" in k:
        continue
    r = b[k] / a[k]
    higher_better = any(s in k for s in ("per_s", "value"))
    worse = (r < 1 - tol) if higher_better else (r > 1 + tol)
    better = (r > 1 + tol) if higher_better else (r < 1 - tol)
    if worse or better:
        print("%-8s %-70s %12.5g -> %12.5g  (x%.3f)" % ("WORSE" if worse else "better", k, a[k], b[k], r))
        bad += worse
print("%d fields compared, %d worse by more than %.0f %%" % (len(set(a) & set(b)), bad, tol * 100))
sys.exit(1 if bad else 0)
