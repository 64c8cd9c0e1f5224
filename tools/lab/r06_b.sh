cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_b
( time timeout 1500 python -m pytest tests/test_gpu_exact.py::test_exact_mode_inversion_vs_reference tests/test_gpu_exact.py::test_exact_mode_inversion_window_vs_reference tests/test_swan_geometry.py -m gpu -q -s --durations=10 > gpurun_out/r06_b/pytest_new1.log 2>&1 ) 2> gpurun_out/r06_b/t1.txt
tail -25 gpurun_out/r06_b/pytest_new1.log
( time timeout 1500 python -m pytest tests/test_gpu_c2_window.py::test_two_ranks_full_size_sd_windows tests/test_gpu_c3_window.py::test_two_ranks_full_size_svd_windows -m gpu -q -s --durations=10 > gpurun_out/r06_b/pytest_new2.log 2>&1 ) 2> gpurun_out/r06_b/t2.txt
tail -25 gpurun_out/r06_b/pytest_new2.log
