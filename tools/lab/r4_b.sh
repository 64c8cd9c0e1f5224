cd $GRAFT_REPO_ROOT
T=${1:-r04_b}
mkdir -p gpurun_out/$T
timeout 900 python tools/p7x_bench.py > gpurun_out/$T/p7x_bench.txt 2>&1
tail -70 gpurun_out/$T/p7x_bench.txt
VIDSEG_GEMM=p7x=0 VIDSEG_BENCH_PMC=0 timeout 1200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/$T/bench_p7.json 2> gpurun_out/$T/bench_p7.err
tail -12 gpurun_out/$T/bench_p7.err
VIDSEG_BENCH_PMC=0 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/$T/bench_p7x.json 2> gpurun_out/$T/bench_p7x.err
tail -5 gpurun_out/$T/bench_p7x.err
python - <<PY
import json
for f in ("bench_p7", "bench_p7x"):
    try:
        d = json.loads(open("gpurun_out/$T/%s.json" % f).read().strip().splitlines()[-1])
        print(f, "value", d["value"], d["ms_per_step"], d.get("mask_iou_vs_reference", {}).get("mean_iou"), d.get("mask_iou_vs_reference", {}).get("windows_at_0.99"))
        print("  family", json.dumps(d["roofline"]["family"]["by_kernel"]))
        for k in ("two_lanes", "chained_window", "full_schedule", "fast_mode"):
            v = d.get(k) or {}
            print("  ", k, v.get("value"), v.get("ms_per_step"), v.get("mask_iou_vs_reference"), v.get("error"))
        s = d.get("secondary", {}); print("  secondary", s.get("value"), s.get("ms_per_step"), s.get("fast_mode"), s.get("error"), s.get("step4_latent_blending"))
    except Exception as e:
        print(f, "parse failed", e)
PY
