# kernel-trace duration + PMC passes over the isolated fused PE kernel (separate rocprofv3 runs, kernel-trace only)
# usage: bash tools/pmc_pe.sh [M] ; honours MV2D_HIP_LIB
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-8794}
export MV2D_PE_BENCH_MODES=hot
rm -rf $R/gpurun_out/pmcpe
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pmcpe/kt -o p -- python $R/tools/microbench_pe.py $M > /dev/null 2>&1
python $R/tools/rocpd_stats.py $(ls $R/gpurun_out/pmcpe/kt/*/p_results.db 2>/dev/null || ls $R/gpurun_out/pmcpe/kt/p_results.db) 2>/dev/null | grep -i "pe_fused" | head -4
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcpe/p$i -o p -- python $R/tools/microbench_pe.py $M > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(ls $R/gpurun_out/pmcpe/p$i/*/p_results.db 2>/dev/null || ls $R/gpurun_out/pmcpe/p$i/p_results.db) 2>/dev/null | grep -i "pe_fused\|kernel " | head -8
done
