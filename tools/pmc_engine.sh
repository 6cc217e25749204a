#!/bin/bash
# PMC passes over tools/run_engine.py (one stream, eager launches of one 8-sample frame set): HBM-side traffic + L2 hit / request counters per kernel.
# usage: tools/pmc_engine.sh <outdir-under-gpurun_out> [run_engine args...]     (separate --pmc passes with --kernel-trace only)
OUT=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  tag=$(echo $c | tr ' ' '+')
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$OUT/raw -o p -- python tools/run_engine.py --eager --steps 3 "$@" > /dev/null 2>&1 )
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/$OUT/raw -name "p_results.db" | head -1) > $R/gpurun_out/$OUT/pmc_${tag}.txt 2>&1
  rm -rf $R/gpurun_out/$OUT/raw
done
head -12 $R/gpurun_out/$OUT/pmc_FETCH_SIZE.txt
