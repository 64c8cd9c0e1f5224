# full GPU record of a build: bash tools/lab/r3_full.sh <tag>   (GPU suite, headline bench, rocprofv3 kernel stats of the bench and of the exact mode)
cd $GRAFT_REPO_ROOT
T=${1:-r03_c}
mkdir -p gpurun_out/$T
( time timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/$T/pytest_gpu.log 2>&1 ) 2> gpurun_out/$T/pytest_time.txt
tail -5 gpurun_out/$T/pytest_gpu.log; cat gpurun_out/$T/pytest_time.txt | tail -3
( time VIDSEG_ACT=bf16 timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/$T/pytest_gpu_bf16.log 2>&1 ) 2>> gpurun_out/$T/pytest_time.txt
tail -2 gpurun_out/$T/pytest_gpu_bf16.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
tail -c 800 gpurun_out/$T/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_under_rocprof.json 2>/tmp/prof_d.err
db=$(find /tmp/prof_d -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
head -24 $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_x -o r -- python $GRAFT_REPO_ROOT/tools/exact_study.py --windows 0-3 > $GRAFT_REPO_ROOT/gpurun_out/$T/exact_under_rocprof.log 2>/tmp/prof_x.err
db=$(find /tmp/prof_x -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python tools/exact_study.py --windows 0-3 (exact precision mode, 4 windows) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/exact_kernel_stats.md
head -30 $GRAFT_REPO_ROOT/gpurun_out/$T/exact_kernel_stats.md
