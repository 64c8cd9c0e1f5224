# N repeats of the headline bench, per-window IoUs and (VIDSEG_DEBUG_HASH) feature / mask hashes of every analysed window:
#   PROF_EXT=0 REPS=10 bash tools/lab/rep_bench.sh      -> gpurun_out/rep/run_*.txt; differing hashes are reported at the end
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rep
for i in $(seq 1 ${REPS:-6}); do
VIDSEG_DEBUG_HASH=1 VIDSEG_GEMM=ext=${PROF_EXT:-1} VIDSEG_BENCH_EXACT=0 VIDSEG_BENCH_PMC=0 python bench.py --steps 16 --warmup 4 --no-secondary --no-cpu-baseline 2> gpurun_out/rep/err_$i.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mask_iou_vs_reference']
print(d['value'], m['mean_iou'])"
grep HASH gpurun_out/rep/err_$i.txt | sort | uniq -c | awk '{print $3, $5, $7, "x"$1}' > gpurun_out/rep/run_$i.txt
done
cat gpurun_out/rep/run_*.txt | awk '{print $1, $2, $3}' | sort | uniq -c | awk '{print $2}' | sort | uniq -c | awk '$1 > 1 {print "window", $2, "has", $1, "distinct (features, masks) results"}'
for i in $(seq 2 ${REPS:-6}); do diff <(awk '{print $1,$2,$3}' gpurun_out/rep/run_1.txt) <(awk '{print $1,$2,$3}' gpurun_out/rep/run_$i.txt) > /dev/null || { echo "run $i differs from run 1:"; diff <(awk '{print $1,$2,$3}' gpurun_out/rep/run_1.txt) <(awk '{print $1,$2,$3}' gpurun_out/rep/run_$i.txt); }; done
