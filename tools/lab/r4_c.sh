cd $GRAFT_REPO_ROOT
T=${1:-r04_c}
mkdir -p gpurun_out/$T
export AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 VIDSEG_BENCH_PMC=0
timeout 600 python -X faulthandler bench.py --config svd --narrow --steps 1 --warmup 0 --no-secondary --no-cpu-baseline > gpurun_out/$T/svd_narrow.json 2> gpurun_out/$T/svd_narrow.err
echo "narrow rc=$?"; tail -40 gpurun_out/$T/svd_narrow.err | cut -c1-200
timeout 900 python -X faulthandler bench.py --config svd --steps 1 --warmup 0 --no-secondary --no-cpu-baseline > gpurun_out/$T/svd_full.json 2> gpurun_out/$T/svd_full.err
echo "full rc=$?"; tail -40 gpurun_out/$T/svd_full.err | cut -c1-200
