mkdir -p gpurun_out
VIDSEG_GEMM_SHAPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap 2> gpurun_out/shapes_auto.log | tail -1 | cut -c1-120
VIDSEG_GEMM_DMA=3 VIDSEG_GEMM_SHAPES=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap 2> gpurun_out/shapes_d4.log | tail -1 | cut -c1-120
for t in auto d4; do python tools/dbg/shape_summary.py gpurun_out/shapes_$t.log > gpurun_out/shapes_$t.txt; done
