cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof_k
python bench.py > gpurun_out/prof_k/bench.json 2> gpurun_out/prof_k/bench.err
python bench.py --no-overlap --no-cpu-baseline > gpurun_out/prof_k/bench_no_overlap.json 2>> gpurun_out/prof_k/bench.err
python bench.py --masks-only --no-cpu-baseline > gpurun_out/prof_k/bench_masks_only.json 2>> gpurun_out/prof_k/bench.err
python bench.py --vae --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_k/bench_vae.json 2>> gpurun_out/prof_k/bench.err
python bench.py --config svd --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_k/bench_svd.json 2>> gpurun_out/prof_k/bench.err
python bench.py --config svd --fp8-attn --masks 50 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_k/bench_svd_fp8_k50.json 2>> gpurun_out/prof_k/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k/kt -o k -- python bench.py --no-cpu-baseline > gpurun_out/prof_k/bench_under_rocprof.log 2>&1
db=$(find gpurun_out/prof_k/kt -name "*.db" | head -1)
python tools/prof_summary.py $db "r01_k: python bench.py --no-cpu-baseline under rocprofv3 --kernel-trace --stats (final round-1 build)" > gpurun_out/prof_k/kernel_stats.md
rm -rf gpurun_out/prof_k/kt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/prof_k/A -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_k/B -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
python tools/pmc_traffic.py $(find gpurun_out/prof_k/A -name "*.db" | head -1) $(find gpurun_out/prof_k/B -name "*.db" | head -1) gpurun_out/prof_k/traffic.json > /dev/null 2>&1
rm -rf gpurun_out/prof_k/A gpurun_out/prof_k/B
