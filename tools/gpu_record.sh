# GPU record of a build: bash tools/gpu_record.sh <tag> [tests] [bench] [prof] [svdprof] [full] [final]
#   tests    the exact-mode / analysis / window GPU tests (-x)             full   the whole GPU suite instead
#   bench    python bench.py (default flags + steps 16 / warmup 3)         prof   rocprofv3 kernel stats of the SD parity window
#   svdprof  rocprofv3 kernel stats of the SVD parity window               final  bf16 suite, MFMA-utilisation PMC pass, race stress, determinism
# Everything lands under gpurun_out/<tag>/; the summaries worth keeping are copied to profiles/ by hand.
cd $GRAFT_REPO_ROOT
T=${1:-rec}
shift
mkdir -p gpurun_out/$T
has() { for a in "$@"; do :; done; case " $ARGS " in *" $1 "*) return 0;; esac; return 1; }
ARGS="$*"
if has full; then
  ( time timeout 3000 python -m pytest tests -m gpu -q -s --durations=25 > gpurun_out/$T/pytest_gpu.log 2>&1 ) 2> gpurun_out/$T/pytest_time.txt
  tail -5 gpurun_out/$T/pytest_gpu.log; tail -3 gpurun_out/$T/pytest_time.txt
elif has tests; then
  ( time timeout 1500 python -m pytest tests/test_gpu_exact.py tests/test_gpu_analysis.py tests/test_gpu_c2_window.py tests/test_gpu_c3_window.py -m gpu -q -s -x > gpurun_out/$T/pytest_gpu.log 2>&1 ) 2> gpurun_out/$T/pytest_time.txt
  tail -5 gpurun_out/$T/pytest_gpu.log; tail -3 gpurun_out/$T/pytest_time.txt
fi
grep -E "windows at IoU|reproduce the reference|window [0-9]+ (exact|parity|fp16)|nrms|max err" gpurun_out/$T/pytest_gpu.log 2>/dev/null | tail -40
if has bench; then
  ( time timeout 1200 python bench.py > gpurun_out/$T/bench_default.json 2> gpurun_out/$T/bench_default.err ) 2> gpurun_out/$T/bench_default_time.txt; tail -3 gpurun_out/$T/bench_default_time.txt
  cp bench_full.json gpurun_out/$T/bench_default_full.json; wc -c gpurun_out/$T/bench_default.json
  timeout 1800 python bench.py --steps 16 --warmup 3 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
  tail -c 400 gpurun_out/$T/bench.err
  cp bench_full.json gpurun_out/$T/bench_full.json; wc -c gpurun_out/$T/bench.json
  python tools/bench_digest.py gpurun_out/$T/bench_full.json
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  rm -rf /tmp/prof_d
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --lanes 1 > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_under_rocprof.json 2>/tmp/prof_d.err
  db=$(find /tmp/prof_d -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --lanes 1 (parity mode) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
  head -34 $GRAFT_REPO_ROOT/gpurun_out/$T/bench_kernel_stats.md
fi
if has svdprof; then
  rm -rf /tmp/prof_s
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o r -- python $GRAFT_REPO_ROOT/bench.py --config svd --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-step4 --lanes 1 > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_svd_under_rocprof.json 2>/tmp/prof_s.err
  db=$(find /tmp/prof_s -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python bench.py --config svd --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-step4 --lanes 1 (parity mode) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/bench_svd_kernel_stats.md
  head -44 $GRAFT_REPO_ROOT/gpurun_out/$T/bench_svd_kernel_stats.md
  tail -c 300 /tmp/prof_s.err
fi
if has final; then
  cd $GRAFT_REPO_ROOT
  ( time VIDSEG_DIST_BACKEND=gloo VIDSEG_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/$T/bench_2ranks_one_gpu.json 2> gpurun_out/$T/bench_2ranks_one_gpu.err ) 2>> gpurun_out/$T/pytest_time.txt
  cp bench_full.json gpurun_out/$T/bench_2ranks_one_gpu_full.json; wc -c gpurun_out/$T/bench_2ranks_one_gpu.json
  ( time VIDSEG_ACT=bf16 timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/$T/pytest_gpu_bf16.log 2>&1 ) 2>> gpurun_out/$T/pytest_time.txt
  tail -2 gpurun_out/$T/pytest_gpu_bf16.log
  cd /tmp
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-secondary > /tmp/pm.log 2>&1
  db=$(find /tmp/pm -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_mfma_util.py $db $GRAFT_REPO_ROOT/gpurun_out/$T/mfma_util.json | tail -24
  rm -rf /tmp/prof_s45; timeout 800 rocprofv3 --kernel-trace --stats -d /tmp/prof_s45 -o r -- python $GRAFT_REPO_ROOT/tools/step45_timing.py > $GRAFT_REPO_ROOT/gpurun_out/$T/step45_under_rocprof.json 2>/tmp/prof_s45.err
  db=$(find /tmp/prof_s45 -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "$T: python tools/step45_timing.py (Steps 1-3 twice, warm-up, 6 passes as round 5 ran them, 40 passes of the sweep, 40 decodes + difference maps + arg-max) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/$T/step45_kernel_stats.md
  head -14 $GRAFT_REPO_ROOT/gpurun_out/$T/step45_kernel_stats.md | cut -c1-160
  rm -rf /tmp/pt; timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum -d /tmp/pt -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-secondary > /tmp/pt.log 2>&1
  db=$(find /tmp/pt -name "*results.db" | head -1)
  python $GRAFT_REPO_ROOT/tools/pmc_by_kernel.py $db k_gemm > $GRAFT_REPO_ROOT/gpurun_out/$T/gemm_l2_counters.md; head -12 $GRAFT_REPO_ROOT/gpurun_out/$T/gemm_l2_counters.md
  cd $GRAFT_REPO_ROOT
  timeout 900 python tools/race_stress.py --exact --twin --iters 60 > gpurun_out/$T/race_stress_exact_twin.txt 2>&1; tail -2 gpurun_out/$T/race_stress_exact_twin.txt
  timeout 1200 python tools/race_stress.py --exact --unet-bg --iters 150 > gpurun_out/$T/race_stress_exact_unet_bg.txt 2>&1; tail -2 gpurun_out/$T/race_stress_exact_unet_bg.txt
  bash tools/lanes_determinism.sh > gpurun_out/$T/lanes_determinism.txt 2>&1; tail -3 gpurun_out/$T/lanes_determinism.txt
  timeout 900 python tools/race_stress.py --exact --iters 100 > gpurun_out/$T/race_stress_exact.txt 2>&1; tail -4 gpurun_out/$T/race_stress_exact.txt
  timeout 600 python tools/determinism_check.py --precision exact --overlap --lanes 2 --masks-only --windows 0-5 --reps 2 > gpurun_out/$T/determinism_exact_a.txt 2>&1
  timeout 600 python tools/determinism_check.py --precision exact --overlap --lanes 2 --masks-only --windows 0-5 --reps 2 > gpurun_out/$T/determinism_exact_b.txt 2>&1
  diff <(grep -v "^\[" gpurun_out/$T/determinism_exact_a.txt) <(grep -v "^\[" gpurun_out/$T/determinism_exact_b.txt) > gpurun_out/$T/determinism_exact_diff.txt && echo "determinism: two processes identical" || (echo "determinism: DIFF"; head -5 gpurun_out/$T/determinism_exact_diff.txt)
  tail -3 gpurun_out/$T/determinism_exact_a.txt
fi
