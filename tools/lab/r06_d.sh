cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_d
( time timeout 1200 python -m pytest tests/test_gpu_exact.py tests/test_gpu_unet.py tests/test_gpu_c2_window.py::test_specified_inputs_windows tests/test_gpu_c3_window.py::test_step4_latent_blending_full_size -m gpu -q -s -k "shared_prefix or step4 or specified or modulat" --durations=8 > gpurun_out/r06_d/pytest.log 2>&1 ) 2> gpurun_out/r06_d/t1.txt
grep -E "specified inputs|passed|failed|Error|error" gpurun_out/r06_d/pytest.log | tail -30
timeout 600 python tools/step45_timing.py > gpurun_out/r06_d/step45.json 2> gpurun_out/r06_d/step45.err; cat gpurun_out/r06_d/step45.json | head -20
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_v; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_v -o r -- python $GRAFT_REPO_ROOT/tools/step45_timing.py --decode-only > $GRAFT_REPO_ROOT/gpurun_out/r06_d/decode_under_rocprof.json 2>/tmp/prof_v.err
db=$(find /tmp/prof_v -name "*results.db" | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db "r06_d: python tools/step45_timing.py --decode-only (6 first-stage decodes of a 14-frame 512x512 window) under rocprofv3 --kernel-trace --stats" > $GRAFT_REPO_ROOT/gpurun_out/r06_d/decode_kernel_stats.md
head -40 $GRAFT_REPO_ROOT/gpurun_out/r06_d/decode_kernel_stats.md
rocprofv3 -L 2>/dev/null | grep -i -E "mall|dram|TCC_EA0|TCC_HIT|TCC_MISS|TCC_REQ\b|TCC_READ|TCC_TAG" | head -60 > $GRAFT_REPO_ROOT/gpurun_out/r06_d/counters_tcc.txt; wc -l $GRAFT_REPO_ROOT/gpurun_out/r06_d/counters_tcc.txt
