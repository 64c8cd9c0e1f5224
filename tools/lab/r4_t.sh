cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_t
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "geglu" 2>&1 | tail -3
for tile in old p7g; do
VIDSEG_GEGLU_TILE=$tile VIDSEG_BENCH_PMC=0 VIDSEG_BENCH_MODES=0 timeout 600 python bench.py --precision fp16 --steps 16 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/r04_t/bench_fp16_$tile.json 2> gpurun_out/r04_t/bench_fp16_$tile.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r04_t/bench_fp16_$tile.json").read().strip().splitlines()[-1])
m = d.get("mask_iou_vs_reference", {})
print("GEGLU tile $tile: fp16 mode value", d["value"], d["ms_per_step"], m.get("mean_iou"), m.get("windows_at_0.99"))
print("  ", json.dumps({k.split(" (")[0]: v for k, v in d["roofline"]["family"]["by_kernel"].items()}))
PY
done
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_c2_window.py -m gpu -q -x 2>&1 | tail -3
