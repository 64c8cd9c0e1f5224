"""Durations of the kernels whose name contains argv[2], grouped by (grid, workgroup) size, from a rocprofv3 --kernel-trace db."""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if "kernel_dispatch" in t and "rocpd" in t] or [t for t in tabs if "kernel" in t]
cols = {t: [c[1] for c in db.execute(f"pragma table_info({t})")] for t in kd}
if len(sys.argv) < 3:
    print(tabs); print(cols); sys.exit()
view = "kernels" if "kernels" in tabs else kd[0]
c = [x[1] for x in db.execute(f"pragma table_info({view})")]
name_c = "name" if "name" in c else "kernel_name"
gx = [x for x in c if x.lower() in ("grid_x", "grid_size_x", "grid_size")][0]
wx = [x for x in c if x.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")][0]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, g, w, s, e in db.execute(f"select {name_c}, {gx}, {wx}, start, end from {view}"):
    if sys.argv[2] in n:
        a = agg[(n.split("(")[0][:40], g, w)]; a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:42s} grid {k[1]:>9} wg {k[2]:>4}  n={n:5d}  avg {us / n:8.1f} us  total {us / 1e3:8.2f} ms  {100 * us / tot:5.1f}%")
