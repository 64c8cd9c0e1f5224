cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_vae.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['family']['achieved'])"; done
VIDSEG_GEMM_PANEL=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nopanel', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['family']['achieved'])"
mkdir -p gpurun_out/pmc
VIDSEG_GEMM_SHAPES=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc/A -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2> gpurun_out/pmc/shapes.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc/B -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > /dev/null 2>&1
python tools/dbg/pmc_per_shape.py $(find gpurun_out/pmc/A -name "*.db" | head -1) $(find gpurun_out/pmc/B -name "*.db" | head -1) gpurun_out/pmc/shapes.log > gpurun_out/pmc/per_shape4.txt 2>&1
python tools/pmc_traffic.py $(find gpurun_out/pmc/A -name "*.db" | head -1) $(find gpurun_out/pmc/B -name "*.db" | head -1) gpurun_out/pmc/traffic4.json > /dev/null 2>&1
rm -rf gpurun_out/pmc/A gpurun_out/pmc/B
