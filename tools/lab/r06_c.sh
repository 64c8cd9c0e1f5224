cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_c
( time timeout 900 python -m pytest tests/test_gpu_exact.py -m gpu -q -s -k "shared_prefix or step4" --durations=5 > gpurun_out/r06_c/pytest_share.log 2>&1 ) 2> gpurun_out/r06_c/t1.txt
tail -12 gpurun_out/r06_c/pytest_share.log
( time timeout 900 python bench.py > gpurun_out/r06_c/bench_default.json 2> gpurun_out/r06_c/bench_default.err ) 2> gpurun_out/r06_c/bench_default_time.txt
cp bench_full.json gpurun_out/r06_c/bench_default_full.json
wc -c gpurun_out/r06_c/bench_default.json; tail -3 gpurun_out/r06_c/bench_default_time.txt; tail -4 gpurun_out/r06_c/bench_default.err
python -c "
import json;d=json.load(open('gpurun_out/r06_c/bench_default_full.json'));print(json.dumps(d.get('step45'),indent=1));print(json.dumps(d['secondary'].get('step4_latent_blending'),indent=1)[:1200])"
