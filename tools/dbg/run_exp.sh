cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_svd
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_svd/kt -o s -- python bench.py --config svd --steps 2 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/prof_svd/log.txt 2>&1
db=$(find gpurun_out/prof_svd/kt -name "*.db" | head -1)
python tools/prof_summary.py $db "r01_i: bench.py --config svd --steps 2 --warmup 1 --no-overlap under rocprofv3" > gpurun_out/prof_svd/kernel_stats.md
rm -rf gpurun_out/prof_svd/kt
