# end-of-round GPU record: bash tools/r4_final.sh <tag>   (fp16 + bf16 GPU suites, default bench, rocprofv3 kernel stats, MFMA-utilisation PMC pass)
cd $GRAFT_REPO_ROOT
T=${1:-r04_f}
mkdir -p gpurun_out/$T
( time timeout 3000 python -m pytest tests -m gpu -q -s > gpurun_out/$T/pytest_gpu.log 2>&1 ) 2> gpurun_out/$T/pytest_time.txt
tail -4 gpurun_out/$T/pytest_gpu.log; tail -3 gpurun_out/$T/pytest_time.txt
grep -E "reproduce the reference|configs\[4\] window|window [0-9]+ (exact|parity|fp16)" gpurun_out/$T/pytest_gpu.log | tail -24
( time VIDSEG_ACT=bf16 timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/$T/pytest_gpu_bf16.log 2>&1 ) 2>> gpurun_out/$T/pytest_time.txt
tail -2 gpurun_out/$T/pytest_gpu_bf16.log
timeout 1800 python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
tail -4 gpurun_out/$T/bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$T/bench.json").read().strip().splitlines()[-1])
    m = d.get("mask_iou_vs_reference", {})
    print("value", d["value"], d["ms_per_step"], m.get("mean_iou"), m.get("windows_at_0.99"), m.get("n_windows"))
    for k in ("two_lanes", "chained_window", "full_schedule", "fast_mode"):
        v = d.get(k) or {}
        print(k, v.get("value"), v.get("ms_per_step"), v.get("mask_iou_vs_reference"), v.get("error"))
    r = d["roofline"]; print("roofline", r["kernel"][:20], r["achieved"], r["frac"], r["avg_launch_us"], r.get("traffic"), r.get("algorithmic_bytes"))
    s = d.get("secondary", {}); print("secondary", s.get("value"), s.get("ms_per_step"), s.get("mask_iou_vs_reference", {}).get("windows") if s.get("mask_iou_vs_reference") else None, s.get("fast_mode"), s.get("error"))
    s = d.get("secondary_fp8", {}); print("secondary_fp8", s.get("value"), s.get("ms_per_step"), s.get("roofline"), s.get("error"))
    print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
except Exception as e:
    print("bench parse failed", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_under_rocprof.json 2>/tmp/prof_d.err
db=$(find /tmp/prof_d -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary (parity mode) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
head -16 $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-secondary > /tmp/pm.log 2>&1
db=$(find /tmp/pm -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/pmc_mfma_util.py $db $GRAFT_REPO_ROOT/gpurun_out/$T/mfma_util.json | tail -24
cd $GRAFT_REPO_ROOT
timeout 900 python tools/race_stress.py --exact --iters 100 > gpurun_out/$T/race_stress_exact.txt 2>&1; tail -4 gpurun_out/$T/race_stress_exact.txt
timeout 600 python tools/determinism_check.py --precision exact --overlap --windows 0-5 --reps 2 > gpurun_out/$T/determinism_exact_a.txt 2>&1
timeout 600 python tools/determinism_check.py --precision exact --overlap --windows 0-5 --reps 2 > gpurun_out/$T/determinism_exact_b.txt 2>&1
diff <(grep -v "^\[" gpurun_out/$T/determinism_exact_a.txt) <(grep -v "^\[" gpurun_out/$T/determinism_exact_b.txt) > gpurun_out/$T/determinism_exact_diff.txt && echo "determinism: two processes identical" || (echo "determinism: DIFF"; head -5 gpurun_out/$T/determinism_exact_diff.txt)
tail -3 gpurun_out/$T/determinism_exact_a.txt
