cd $GRAFT_REPO_ROOT
T=${1:-r03_a}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gpu_parallel.py -x -q > gpurun_out/$T/pytest_parallel.log 2>&1; tail -3 gpurun_out/$T/pytest_parallel.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
tail -c 1500 gpurun_out/$T/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_under_rocprof.json 2>/tmp/prof_d.err
db=$(find /tmp/prof_d -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/kernel_stats.md
head -40 $GRAFT_REPO_ROOT/gpurun_out/$T/kernel_stats.md
