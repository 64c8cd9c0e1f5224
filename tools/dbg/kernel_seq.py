"""Durations (us) of every dispatch of kernels whose name contains <substr>, in order: python tools/dbg/kernel_seq.py <db> <substr> [max]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel_dispatch" in t]
rows = list(db.execute(f"select name, start, end from {kt[0]} order by start"))
sel = [(e - s) / 1e3 for n, s, e in rows if sys.argv[2] in n]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 400
print(sys.argv[2], len(sel), "dispatches; last", mx)
print(" ".join(f"{v:.0f}" for v in sel[-mx:]))
