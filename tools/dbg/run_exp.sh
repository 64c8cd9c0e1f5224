cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d gpurun_out/pmc/U -o p -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap > gpurun_out/pmc/util.log 2>&1
db=$(find gpurun_out/pmc/U -name "*.db" | head -1)
python - <<PY
import sqlite3
db=sqlite3.connect("$db")
print([r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")][:40])
print([r[1] for r in db.execute("pragma table_info(counters_collection)")])
PY
python tools/pmc_mfma_util.py $db gpurun_out/pmc/mfma_util.json 2>&1 | tail -12
cp $db gpurun_out/pmc/util.db; rm -rf gpurun_out/pmc/U
