cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_h
# headline runs (default build, fp16): overlapped and not
python bench.py > gpurun_out/prof_h/bench.json 2> gpurun_out/prof_h/bench.err
python bench.py --no-overlap --no-cpu-baseline > gpurun_out/prof_h/bench_no_overlap.json 2>> gpurun_out/prof_h/bench.err
python bench.py --config svd --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_h/bench_svd.json 2>> gpurun_out/prof_h/bench.err
python bench.py --config svd --fp8-attn --masks 50 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_h/bench_svd_fp8_k50.json 2>> gpurun_out/prof_h/bench.err
# kernel trace of the same command as the headline line (default flags except the host-side legs)
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_h/kt -o h -- python bench.py --no-cpu-baseline > gpurun_out/prof_h/bench_under_rocprof.log 2>&1
db=$(find gpurun_out/prof_h/kt -name "*.db" | head -1)
python tools/prof_summary.py $db "r01_h: python bench.py --no-cpu-baseline under rocprofv3 --kernel-trace --stats (fp16 build, phased big tile, window pipeline on)" > gpurun_out/prof_h/kernel_stats.md
rm -rf gpurun_out/prof_h/kt
# PMC passes (separate runs, kernel-trace only)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_h/A -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_h/B -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
python tools/pmc_traffic.py $(find gpurun_out/prof_h/A -name "*.db" | head -1) $(find gpurun_out/prof_h/B -name "*.db" | head -1) gpurun_out/prof_h/traffic.json
rm -rf gpurun_out/prof_h/A gpurun_out/prof_h/B
ls -la gpurun_out/prof_h
