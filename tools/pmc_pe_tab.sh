# kernel-trace duration + SQ counter passes over the isolated PE table kernel at the headline size (70 349 rows = 8 samples of cfg2_s);
# separate rocprofv3 runs, kernel-trace only.  usage: bash tools/pmc_pe_tab.sh [outdir-under-gpurun_out]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmcpetab}
export MV2D_PE_ONLY=96
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p -- python $R/tools/time_pe_tab.py > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(find $OUT/kt -name "p_results.db" | head -1) 2>/dev/null | grep -i "pe_tab\|kernel " | head -3 | tee $OUT/summary.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- python $R/tools/time_pe_tab.py > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $OUT/p$i -name "p_results.db" | head -1) 2>/dev/null | grep -i "pe_tab\|kernel " | head -8 | tee -a $OUT/summary.txt
  rm -rf $OUT/p$i
done
rm -rf $OUT/kt
