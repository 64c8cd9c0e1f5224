"""Average per-dispatch PMC values (and the dispatch duration of the same pass) of the kernels whose name contains argv[2] from a
rocprofv3 results db: python tools/lab/pmc_dump.py <db> <substr>"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
acc = collections.defaultdict(lambda: [0, 0.0])
dur = [0, 0.0]
for name, cn, val, d in db.execute("select kernel_name, counter_name, value, duration from counters_collection"):
    if sys.argv[2] in name:
        a = acc[cn]; a[0] += 1; a[1] += float(val)
        dur[0] += 1; dur[1] += float(d or 0)
for cn, (n, v) in sorted(acc.items()):
    print(f"{cn:40s} {v / n:16.1f}  (n={n})")
if dur[0]:
    print(f"{'dispatch duration (us, this pass)':40s} {dur[1] / dur[0] / 1e3:16.1f}")
