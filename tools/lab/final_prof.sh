cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_d
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_d/bench.json 2> gpurun_out/r02_d/bench.err
tail -c 600 gpurun_out/r02_d/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/r02_d/bench_under_rocprof.json 2>/tmp/prof_d.err
db=$(find /tmp/prof_d -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "r02_d: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/r02_d/kernel_stats.md
head -30 $GRAFT_REPO_ROOT/gpurun_out/r02_d/kernel_stats.md
