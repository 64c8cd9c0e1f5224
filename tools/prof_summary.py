"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: python tools/prof_summary.py <results.db> [title]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print(f"# {title}\n")
print("rocprofv3 --kernel-trace --stats (durations in microseconds, whole process incl. warm-up)\n")
print("| kernel | calls | total us | avg us | % |")
print("|---|---|---|---|---|")
for n, c, t, a, p in rows[:48]:
    print(f"| `{n[:260] if 'at::native' in n else n[:90]}` | {c} | {t:.0f} | {a:.1f} | {p:.2f} |")
print(f"\ntotal kernel time: {sum(r[2] for r in rows):.0f} us over {sum(r[1] for r in rows)} dispatches")
