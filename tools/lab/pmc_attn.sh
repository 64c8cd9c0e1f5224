# PMC passes over tools/lab/attn_bench.py (GPU box): bash tools/lab/pmc_attn.sh [kernel substring]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-k_attention3}
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_LDS" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/pa$i -o r -- python $R/tools/lab/attn_bench.py > /tmp/pa$i.log 2>&1
  db=$(find /tmp/pa$i -name "*results.db" | head -1)
  echo "== $set"; python $R/tools/lab/pmc_dump.py $db $K 2>&1 | tail -8
done
