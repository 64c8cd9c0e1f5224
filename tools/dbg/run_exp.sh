timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py tests/test_gpu_vae.py -q -m gpu -x 2>&1 | tail -3
VIDSEG_GEMM_BIG=2 timeout 120 python tools/dbg/ph_ksweep.py 2>&1 | grep "^M" | head -6
for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['family']['achieved'], {k: v['tflops'] for k, v in d['roofline']['family']['by_kernel'].items()})"; done
