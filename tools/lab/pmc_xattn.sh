# PMC passes over tools/xattn_bench.py --first (GPU box): bash tools/lab/pmc_xattn.sh [kernel substring]   (VIDSEG_ATTN passes through)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=${1:-k_x_attention_mfma}
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_ACTIVE_INST_FLAT SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/px$i
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/px$i -o r -- python $R/tools/xattn_bench.py --first --reps 5 > /tmp/px$i.log 2>&1
  db=$(find /tmp/px$i -name "*results.db" | head -1)
  echo "== $set"; python $R/tools/lab/pmc_dump.py $db $K 2>&1 | tail -8
done
