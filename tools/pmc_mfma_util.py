"""MFMA utilisation and effective clock of the GEMM kernels from one rocprofv3 --pmc pass:

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d D -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap
    python tools/pmc_mfma_util.py D/.../p_results.db profiles/r01_mfma_util.json

Per kernel name: effective clock = GRBM_GUI_ACTIVE / dispatch duration (MI355X_MICROARCH.md, DVFS give-back);
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 1024 SIMDs) -- GUI_ACTIVE averaged over the 8 XCDs, the MFMA counter counts busy cycles per SIMD
(32 per v_mfma_f32_32x32x16), summed over the chip; utilisation vs the 2.4 GHz peak = that * effective clock / 2.4.
Profiled passes run at slightly lower clocks than un-profiled ones; the utilisation ratio does not depend on it."""
import collections
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection")) if "start" in cols else None
if rows is None:
    dur = {d: (e - s) for d, s, e in db.execute("select dispatch_id, start, end from kernels")} if True else {}
    rows = [(d, n, c, v, 0, dur.get(d, 0)) for d, n, c, v in db.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection")]
per = collections.defaultdict(lambda: collections.defaultdict(float))
inst = collections.defaultdict(lambda: collections.defaultdict(int))      # rows per (dispatch, counter): one per XCD for GRBM counters
names, durs = {}, {}
for d, n, c, v, s, e in rows:
    per[d][c] += float(v)
    inst[d][c] += 1
    names[d] = n.split("(")[0]
    durs[d] = float(e - s)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d, cs in per.items():
    if not any(k in names[d] for k in ("k_gemm", "attention")):
        continue
    a = agg[names[d]]
    a["launches"] += 1
    a["dur_ns"] += durs[d]
    for c, v in cs.items():
        # rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (one GRBM each): / 8 = shader cycles of the dispatch
        a[c] += v / (8.0 if c == "GRBM_GUI_ACTIVE" and inst[d][c] == 1 else float(inst[d][c]) if c == "GRBM_GUI_ACTIVE" else 1.0)
out = {}
for n, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur_ns"]):
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    clk = gui / a["dur_ns"] if a["dur_ns"] else 0.0                       # cycles per ns = GHz
    util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024.0) if gui else 0.0
    wait = a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else None
    out[n] = {"launches": int(a["launches"]), "total_ms": round(a["dur_ns"] / 1e6, 3), "effective_clock_ghz": round(clk, 3),
              "mfma_util_at_effective_clock": round(util, 4), "mfma_util_vs_2p4ghz_peak": round(util * clk / 2.4, 4),
              "wave_parked_fraction": None if wait is None else round(wait, 4)}
    print(n, out[n])
if len(sys.argv) > 2:
    json.dump({"_how": __doc__, "per_kernel": out}, open(sys.argv[2], "w"), indent=1)
