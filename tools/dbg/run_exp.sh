cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
VIDSEG_GEMM_SHAPES=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc/A -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2> gpurun_out/pmc/shapes.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc/B -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
python tools/dbg/pmc_per_shape.py $(find gpurun_out/pmc/A -name "*.db" | head -1) $(find gpurun_out/pmc/B -name "*.db" | head -1) gpurun_out/pmc/shapes.log all > gpurun_out/pmc/per_shape_all.txt 2>&1
rm -rf gpurun_out/pmc/A gpurun_out/pmc/B
