mkdir -p gpurun_out
VIDSEG_GEMM_SHAPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap 2> gpurun_out/shapes_auto.log | tail -1 | cut -c1-120
VIDSEG_GEMM_BIG=2 VIDSEG_GEMM_SHAPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap 2> gpurun_out/shapes_big2.log | tail -1 | cut -c1-120
VIDSEG_GEMM_MID=2 VIDSEG_GEMM_SHAPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap 2> gpurun_out/shapes_mid2.log | tail -1 | cut -c1-120
for t in auto big2 mid2; do python tools/dbg/shape_summary.py gpurun_out/shapes_$t.log > gpurun_out/shapes_$t.txt; done
