#!/usr/bin/env python3
"""usage: tools/cmp_breakdown.py a.log b.log -- per-layer total_ms of two `bench.py --breakdown` logs side by side"""
import sys

def parse(path):
    d, order = {}, []
    for l in open(path, errors="replace"):
        f = l.split()
        if len(f) >= 6 and f[0] not in ("layer",) and not l.startswith(("{", "W2", "E2", "/")):
            try:
                ms = float(f[-3]); n = int(f[-4])
            except ValueError:
                continue
            d[f[0]] = (ms, n); order.append(f[0])
    return d, order

a, order = parse(sys.argv[1]); b, _ = parse(sys.argv[2])
ta = tb = 0.0
for k in order:
    if k in b:
        ta += a[k][0]; tb += b[k][0]
        print("%-24s %9.3f -> %9.3f ms  (%+5.1f %%)   per launch %7.3f -> %7.3f ms" % (k, a[k][0], b[k][0], 100 * (b[k][0] / a[k][0] - 1), a[k][0] / a[k][1], b[k][0] / b[k][1]))
print("%-24s %9.3f -> %9.3f ms  (%+5.1f %%)" % ("sum", ta, tb, 100 * (tb / ta - 1)))
