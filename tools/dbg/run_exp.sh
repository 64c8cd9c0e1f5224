cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_h
cd $R
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_h -o h -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/prof_h/bench_under_rocprof.log 2>&1
db=$(find gpurun_out/prof_h -name "*.db" | head -1)
python tools/prof_summary.py $db "r01_h: bench.py --steps 3 --warmup 1 --no-overlap under rocprofv3 (fp16 build, phased big tile)" > gpurun_out/prof_h/kernel_stats.md
tail -1 gpurun_out/prof_h/bench_under_rocprof.log | cut -c1-300
rm -f $db
