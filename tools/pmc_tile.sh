# SQ counter passes of the eager single-stream bench, reported for the tile cross-attention kernel (the bench's dominant kernel):
# MFMA busy cycles / instruction mix / wait cycles.  Separate rocprofv3 runs, kernel-trace only.  usage: bash tools/pmc_tile.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmctile}
rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o p -- python $R/bench.py --steps 3 --warmup 1 --inflight 1 --rotate 1 --no-extra-legs --no-graph --no-cpu-baseline > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $OUT/p$i -name "p_results.db" | head -1) 2>/dev/null | grep -i "xattn_tile\|^kernel " | head -6 | tee -a $OUT/summary.txt
  rm -rf $OUT/p$i
done
