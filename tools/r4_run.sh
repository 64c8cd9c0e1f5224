# GPU record of a build: bash tools/r4_run.sh <tag> [quick]   (exact-mode GPU tests, default bench, exact-mode GEMM shape table, rocprofv3 kernel stats)
cd $GRAFT_REPO_ROOT
T=${1:-r04_a}
MODE=${2:-quick}
mkdir -p gpurun_out/$T
if [ "$MODE" = "full" ]; then
  ( time timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/$T/pytest_gpu.log 2>&1 ) 2> gpurun_out/$T/pytest_time.txt
else
  ( time timeout 1200 python -m pytest tests/test_gpu_c2_window.py tests/test_gpu_exact.py -m gpu -q -s -x > gpurun_out/$T/pytest_gpu.log 2>&1 ) 2> gpurun_out/$T/pytest_time.txt
fi
tail -5 gpurun_out/$T/pytest_gpu.log; tail -3 gpurun_out/$T/pytest_time.txt
grep -E "exact \+ masks_only|exact mode:" gpurun_out/$T/pytest_gpu.log | tail -4
timeout 1500 python bench.py --steps 16 --warmup 3 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
tail -c 300 gpurun_out/$T/bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], d["ms_per_step"], d.get("mask_iou_vs_reference", {}).get("mean_iou"), d.get("mask_iou_vs_reference", {}).get("windows_at_0.99"))
    for k in ("two_lanes", "chained_window", "full_schedule", "fast_mode"):
        v = d.get(k) or {}
        print(k, v.get("value"), v.get("ms_per_step"), v.get("mask_iou_vs_reference"), v.get("error"))
    r = d["roofline"]; print("roofline", r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r.get("traffic"), r.get("algorithmic_bytes"))
    print("family", json.dumps(r["family"])[:900])
    s = d.get("secondary", {}); print("secondary", s.get("value"), s.get("ms_per_step"), s.get("fast_mode"), s.get("error"), s.get("step4_latent_blending"))
    print("cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench parse failed", e)
PY
VIDSEG_GEMM=shapes=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --one-window > /dev/null 2> gpurun_out/$T/shapes.log
python tools/shape_summary.py gpurun_out/$T/shapes.log > gpurun_out/$T/shape_summary_parity.txt; head -45 gpurun_out/$T/shape_summary_parity.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_under_rocprof.json 2>/tmp/prof_d.err
db=$(find /tmp/prof_d -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary (parity mode) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
head -34 $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
