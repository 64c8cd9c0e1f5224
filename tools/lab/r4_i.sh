cd $GRAFT_REPO_ROOT
T=${1:-r04_i}
mkdir -p gpurun_out/$T
timeout 900 python tools/xsmall_bench.py > gpurun_out/$T/xsmall_bench.txt 2>&1
python - <<PY
import re
rows = {}
for l in open("gpurun_out/$T/xsmall_bench.txt"):
    m = re.match(r"(xsmall[01]) (.*): +([\d.]+) us", l)
    if m: rows.setdefault(m.group(2), {})[m.group(1)] = float(m.group(3))
for k, v in rows.items():
    print(f"{k:60s} {v.get('xsmall0', 0):8.1f} -> {v.get('xsmall1', 0):8.1f} us")
PY
tail -2 gpurun_out/$T/xsmall_bench.txt
timeout 300 python -m pytest tests/test_gpu_exact.py -m gpu -q -x 2>&1 | tail -2
for xs in 0 1; do
VIDSEG_GEMM=xsmall=$xs VIDSEG_BENCH_PMC=0 VIDSEG_BENCH_MODES=0 timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/$T/bench_xs$xs.json 2> gpurun_out/$T/bench_xs$xs.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$T/bench_xs$xs.json").read().strip().splitlines()[-1])
m = d.get("mask_iou_vs_reference", {})
print("xsmall=$xs value", d["value"], d["ms_per_step"], m.get("mean_iou"), m.get("windows_at_0.99"))
print("  ", json.dumps({k.split(" (")[0]: v for k, v in d["roofline"]["family"]["by_kernel"].items()}))
PY
done
