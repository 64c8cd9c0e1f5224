cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_f
( time timeout 2400 python -m pytest tests -m gpu -q -s --durations=30 > gpurun_out/r06_f/pytest_gpu.log 2>&1 ) 2> gpurun_out/r06_f/pytest_time.txt
tail -40 gpurun_out/r06_f/pytest_gpu.log | grep -v "^$" | cut -c1-200; tail -3 gpurun_out/r06_f/pytest_time.txt
timeout 600 python tools/step45_timing.py > gpurun_out/r06_f/step45.json 2> gpurun_out/r06_f/step45.err; head -16 gpurun_out/r06_f/step45.json
