"""python tools/dbg/pmc_dump.py <db> <kernel substr>: mean of every collected counter over the matching dispatches."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select counter_name, avg(value), count(*), avg(end-start) from counters_collection where kernel_name like ? group by counter_name", ("%" + sys.argv[2] + "%",)))
for r in rows:
    print(f"{r[0]:44s} {r[1]:16.1f}  n={r[2]}  dur_ns={r[3]:.0f}")
