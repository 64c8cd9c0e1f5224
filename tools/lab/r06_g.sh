cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s45; timeout 800 rocprofv3 --kernel-trace --stats -d /tmp/prof_s45 -o r -- python $GRAFT_REPO_ROOT/tools/step45_timing.py > $GRAFT_REPO_ROOT/gpurun_out/r06_g_step45_under_rocprof.json 2>/tmp/prof_s45.err
db=$(find /tmp/prof_s45 -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "r06_g: python tools/step45_timing.py (Steps 1-3 twice, warm-up, 6 passes in full, 40 passes with the shared prefix, 40 decodes + difference maps + arg-max) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/r06_g_step45_kernel_stats.md
head -36 $GRAFT_REPO_ROOT/gpurun_out/r06_g_step45_kernel_stats.md | cut -c1-200
head -12 $GRAFT_REPO_ROOT/gpurun_out/r06_g_step45_under_rocprof.json
