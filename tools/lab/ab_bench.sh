# same-box A/B of the headline: bash tools/lab/ab_bench.sh "ENV_A" "ENV_B" [extra bench args]   (env strings like "VIDSEG_SIDE_SKIP=0")
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; shift 2
for rep in 1 2; do
  for cfg in "$A" "$B"; do
    v=$(env $cfg VIDSEG_BENCH_EXACT=0 VIDSEG_BENCH_PMC=0 python bench.py --steps 16 --warmup 4 --no-secondary --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['family']['gemm_ms_per_step'], d['mask_iou_vs_reference']['mean_iou'])")
    echo "[$cfg] frames/s ms/step gemm_ms mean_iou: $v"
  done
done
