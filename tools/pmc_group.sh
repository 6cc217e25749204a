# SQ counter passes for the shared-tile cross attention (xattn_group_kernel) on one engine run: bash tools/pmc_group.sh [workload] [batch]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-cfg3_t}; B=${2:-16}
OUT=$R/gpurun_out/pmcgroup_$W
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- python tools/run_engine.py --workload $W --batch $B --steps 2 --eager --group 1 > /dev/null 2>&1 )
  python $R/tools/rocpd_pmc.py $(find $OUT/p$i -name "p_results.db" | head -1) 2>/dev/null | grep -i "xattn_group_kernel\|^kernel " | head -6 | tee -a $OUT/summary.txt
  rm -rf $OUT/p$i
done
