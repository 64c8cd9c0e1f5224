"""Join per-dispatch PMC values of the big GEMM tile with the shape log of the same run.
usage: python tools/pmc_per_shape.py <FETCH.db> <WRITE.db> <shape log of the FETCH run>"""
import collections
import re
import sqlite3
import sys


ALL = len(sys.argv) > 4 and sys.argv[4] == "all"      # every kernel of the family instead of the big tile only
FAM = ("k_gemm_ph", "k_gemm_p7", "k_gemm_dma", "k_gemm_tile", "k_gemm_conv") if ALL else ("k_gemm_ph", "k_gemm_p7")
KINDS = (0, 1, 2, 3, 4) if ALL else (1, 4)


def vals(db_path, counter):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (counter,)))
    agg = collections.OrderedDict()
    for d, n, v in rows:                      # one row per XCD/instance: sum per dispatch
        if any(f in n for f in FAM):
            agg[d] = agg.get(d, 0.0) + float(v)
    return list(agg.values())


fetch, write = vals(sys.argv[1], "FETCH_SIZE"), vals(sys.argv[2], "WRITE_SIZE")
shapes = []
for line in open(sys.argv[3]):
    m = re.match(r"GEMMSHAPE M=(\d+) N=(\d+) K=(\d+) ks=(\d+) up=(\d+) st=(\d+) act=(\d+) split=(\d+) us=([\d.]+)(?: kind=(\d+))?", line)
    if m and int(m.group(10) or 0) in KINDS:
        shapes.append(tuple(map(int, m.groups()[:8])) + (int(m.group(10) or 0),))
print(len(fetch), len(write), len(shapes))
n = min(len(fetch), len(write), len(shapes))
agg = collections.OrderedDict()
for s, f, w in zip(shapes[:n], fetch[:n], write[:n]):
    a = agg.setdefault(s, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += f
    a[2] += w
print(f"{'M':>7} {'N':>6} {'K':>6} ks up st act sp {'n':>4} {'fetch MB':>9} {'write MB':>9} {'alg MB':>8} {'ratio':>6}")
for (M, N, K, ks, up, st, act, sp, kind), (c, f, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    taps = 9 if ks == 3 else 1
    alg = M * (K // taps) / (up * up) * (st * st) * 2 + N * K * 2 + M * N * 2 / (2 if act == 2 else 1)
    fm, wm = 2 * f * 1024 / c / 1e6, w * 1024 / c / 1e6
    print(f"{M:7d} {N:6d} {K:6d} {ks:2d} {up:2d} {st:2d} {act:3d} {sp:2d} {c:4d} {fm:9.1f} {wm:9.1f} {alg/1e6:8.1f} {(fm+wm)/(alg/1e6):6.2f} kind{kind}")
