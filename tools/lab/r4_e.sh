cd $GRAFT_REPO_ROOT
T=${1:-r04_e}
mkdir -p gpurun_out/$T
timeout 300 python -m pytest tests/test_gpu_exact.py -m gpu -q -x 2>&1 | tail -3
VIDSEG_BENCH_PMC=0 timeout 900 python bench.py --config svd --steps 3 --warmup 1 --no-secondary --no-cpu-baseline > gpurun_out/$T/svd_parity.json 2> gpurun_out/$T/svd_parity.err
tail -3 gpurun_out/$T/svd_parity.err
python - <<PY
import json
d = json.loads(open("gpurun_out/$T/svd_parity.json").read().strip().splitlines()[-1])
print("svd parity value", d["value"], d["ms_per_step"], d.get("mask_iou_vs_reference"))
print(json.dumps(d["roofline"]["family"])[:1200])
PY
cd /tmp && export TMPDIR=/tmp
VIDSEG_BENCH_PMC=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o r -- python $GRAFT_REPO_ROOT/bench.py --config svd --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/$T/svd_under_rocprof.json 2>/tmp/prof_s.err
db=$(find /tmp/prof_s -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python bench.py --config svd --steps 2 --warmup 1 --no-cpu-baseline --no-secondary (SVD configs[2], parity mode) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/svd_kernel_stats.md
head -34 $GRAFT_REPO_ROOT/gpurun_out/$T/svd_kernel_stats.md
