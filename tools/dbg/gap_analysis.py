"""GPU idle-gap analysis of a rocprofv3 rocpd sqlite kernel trace: python tools/dbg/gap_analysis.py <db> [skip_frac]
Looks only at the last `1-skip_frac` of the dispatches (timed region)."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel_dispatch" in t]
cols = [r[1] for r in db.execute(f"pragma table_info({kt[0]})")]
rows = list(db.execute(f"select name, start, end from {kt[0]} order by start"))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = collections.defaultdict(lambda: [0, 0])
kern = collections.defaultdict(lambda: [0, 0])
prev_end, prev_name = rows[0][2], rows[0][0]
for n, s, e in rows[1:]:
    g = max(0, s - prev_end)
    gaps[prev_name[:50] + " -> " + n[:50]][0] += 1
    gaps[prev_name[:50] + " -> " + n[:50]][1] += g
    kern[n[:70]][0] += 1
    kern[n[:70]][1] += e - s
    prev_end, prev_name = max(prev_end, e), n
print(f"dispatches {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms")
print("top gaps (ms total, count, avg us):")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t/1e6:8.2f} {c:6d} {t/c/1e3:8.1f}  {k}")
print("kernels (ms total, count, avg us):")
for k, (c, t) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {t/1e6:8.2f} {c:6d} {t/c/1e3:8.1f}  {k}")
