cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_a
( time timeout 900 python bench.py > gpurun_out/r06_a/bench_default.json 2> gpurun_out/r06_a/bench_default.err ) 2> gpurun_out/r06_a/bench_default_time.txt
cp bench_full.json gpurun_out/r06_a/bench_default_full.json
wc -c gpurun_out/r06_a/bench_default.json; tail -3 gpurun_out/r06_a/bench_default_time.txt
python -c "
import json;d=json.loads(open('gpurun_out/r06_a/bench_default.json').read().strip().splitlines()[-1]);print(d['value'],d['roofline']['frac'],d['cpu_baseline']['value'],d['secondary']['value'])"
( time VIDSEG_DIST_BACKEND=gloo VIDSEG_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r06_a/bench_2ranks.json 2> gpurun_out/r06_a/bench_2ranks.err ) 2> gpurun_out/r06_a/bench_2ranks_time.txt
cp bench_full.json gpurun_out/r06_a/bench_2ranks_full.json
wc -c gpurun_out/r06_a/bench_2ranks.json; tail -3 gpurun_out/r06_a/bench_2ranks_time.txt; tail -5 gpurun_out/r06_a/bench_2ranks.err
cat gpurun_out/r06_a/bench_2ranks.json | tail -1 | cut -c1-3000
