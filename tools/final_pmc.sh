# MFMA utilisation / effective clock of the final build (one PMC pass over one window): bash tools/final_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/pm -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-overlap --no-secondary > /tmp/pm.log 2>&1
db=$(find /tmp/pm -name "*results.db" | head -1)
mkdir -p $R/gpurun_out/r02_d
python $R/tools/pmc_mfma_util.py $db $R/gpurun_out/r02_d/mfma_util.json | tail -20
