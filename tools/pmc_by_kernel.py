"""Per-kernel averages of the counters of one rocprofv3 --kernel-trace --pmc pass: python tools/pmc_by_kernel.py <results.db> [name filter]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for name, counter, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.split("(")[0].replace("void ", "")
    if flt and flt not in k:
        continue
    a = acc[k][counter]
    a[0] += 1
    a[1] += float(val)
counters = sorted({c for v in acc.values() for c in v})
print("| kernel | launches | " + " | ".join(f"{c} / launch" for c in counters) + " |")
print("|---|---|" + "---|" * len(counters))
for k, v in sorted(acc.items(), key=lambda kv: -max(x[1] for x in kv[1].values())):
    n = max(x[0] for x in v.values())
    print(f"| `{k[:60]}` | {n} | " + " | ".join(f"{v[c][1] / max(v[c][0], 1):.4g}" if c in v else "-" for c in counters) + " |")
