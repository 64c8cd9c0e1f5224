cd $GRAFT_REPO_ROOT
T=${1:-r04_g}
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gpu_exact.py -m gpu -q -x -s 2>&1 | grep -E "fused GEGLU|passed|failed|Error|error" | tail -14
for tile in ph p7x; do
VIDSEG_X_GEGLU_TILE=$tile VIDSEG_BENCH_PMC=0 VIDSEG_BENCH_MODES=0 timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/$T/bench_$tile.json 2> gpurun_out/$T/bench_$tile.err
tail -2 gpurun_out/$T/bench_$tile.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$T/bench_$tile.json").read().strip().splitlines()[-1])
m = d.get("mask_iou_vs_reference", {})
print("$tile value", d["value"], d["ms_per_step"], m.get("mean_iou"), m.get("windows_at_0.99"))
print("  ", json.dumps(d["roofline"]["family"]["by_kernel"]))
PY
done
