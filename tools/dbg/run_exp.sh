cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_i
python bench.py > gpurun_out/prof_i/bench.json 2> gpurun_out/prof_i/bench.err
python bench.py --no-overlap --no-cpu-baseline > gpurun_out/prof_i/bench_no_overlap.json 2>> gpurun_out/prof_i/bench.err
python bench.py --vae --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_i/bench_vae.json 2>> gpurun_out/prof_i/bench.err
python bench.py --config svd --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_i/bench_svd.json 2>> gpurun_out/prof_i/bench.err
python bench.py --config svd --fp8-attn --masks 50 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_i/bench_svd_fp8_k50.json 2>> gpurun_out/prof_i/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_i/kt -o i -- python bench.py --no-cpu-baseline > gpurun_out/prof_i/bench_under_rocprof.log 2>&1
db=$(find gpurun_out/prof_i/kt -name "*.db" | head -1)
python tools/prof_summary.py $db "r01_i: python bench.py --no-cpu-baseline under rocprofv3 --kernel-trace --stats (fp16 build, phased big tile, chunk-major K, window pipeline on)" > gpurun_out/prof_i/kernel_stats.md
rm -rf gpurun_out/prof_i/kt
