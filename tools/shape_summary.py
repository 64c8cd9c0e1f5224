"""Aggregate `GEMMSHAPE` lines (VIDSEG_GEMM=shapes=1 python bench.py ... 2> log) per problem shape."""
import collections, re, sys
agg = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"GEMMSHAPE M=(\d+) N=(\d+) K=(\d+) ks=(\d+) up=(\d+) st=(\d+) act=(\d+) split=(\d+) us=([\d.]+)(?: kind=(\d+))?", line)
    if not m:
        continue
    M, N, K, ks, up, st, act, sp = map(int, m.groups()[:8])
    us = float(m.group(9))
    kind = int(m.group(10) or 0)
    a = agg.setdefault((M, N, K, ks, up, st, act, sp), [0, 0.0, kind])
    a[2] = kind
    a[0] += 1
    a[1] += us
tot = sum(a[1] for a in agg.values())
print(f"{'M':>8} {'N':>6} {'K':>6} ks up st act sp {'n':>5} {'us/launch':>10} {'TF/s':>8} {'share%':>7}")
for k, (n, us, kind) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, ks, up, st, act, sp = k
    print(f"{M:8d} {N:6d} {K:6d} {ks:2d} {up:2d} {st:2d} {act:3d} {sp:2d} {n:5d} {us/n:10.1f} {2.0*M*N*K*n/us/1e6:8.1f} {100*us/tot:7.2f} kind{kind}")
print(f"total {tot/1e3:.2f} ms")
