# PMC passes over the isolated K/V projection kernel (separate rocprofv3 runs, kernel-trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS" "TA_TA_BUSY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmckv/p$i -o p -- python $R/tools/microbench_kv.py > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(ls $R/gpurun_out/pmckv/p$i/*/p_results.db 2>/dev/null || ls $R/gpurun_out/pmckv/p$i/p_results.db) 2>/dev/null | grep -i "kvproj\|kernel " | head -8
done
