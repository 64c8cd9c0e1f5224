# MFMA utilisation / effective clock of the final build (one PMC pass over one window of each precision mode): bash tools/lab/final_pmc.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r03_d}
mkdir -p $R/gpurun_out/$T
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pm -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-secondary > /tmp/pm.log 2>&1
db=$(find /tmp/pm -name "*results.db" | head -1)
python $R/tools/pmc_mfma_util.py $db $R/gpurun_out/$T/mfma_util.json | tail -20
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pmx -o p -- python $R/tools/exact_study.py --windows 0-0 > /tmp/pmx.log 2>&1
db=$(find /tmp/pmx -name "*results.db" | head -1)
python $R/tools/pmc_mfma_util.py $db $R/gpurun_out/$T/mfma_util_exact_mode.json | tail -12
