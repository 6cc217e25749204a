# SQ counter passes over the dense attention block alone (tools/train_dense_time.py): busy / wait cycles, MFMA busy, instruction mix.
# Separate rocprofv3 runs, kernel-trace only.  usage: bash tools/pmc_dense.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmcdense}
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- python $R/tools/train_dense_time.py > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $OUT/p$i -name "p_results.db" | head -1) 2>/dev/null | grep -i "dense_attn\|^kernel " | head -6 | tee -a $OUT/summary.txt
  rm -rf $OUT/p$i
done
