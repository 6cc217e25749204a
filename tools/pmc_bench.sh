# HBM-side counters of the eager single-stream bench (separate --pmc passes, kernel trace only) + kernel-trace stats of the default bench
# usage: tools/pmc_bench.sh [outdir-under-gpurun_out] [bench args, e.g. --workload cfg3_t --batch 16]   (default outdir pmc2, default workload)
# L2=1 adds the L2 passes (TCC hit / miss, TCP->TCC read requests); STATS=0 skips the kernel-trace summary of the graph-replayed bench.
OUT=${1:-pmc2}
shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
PASSES=("FETCH_SIZE" "WRITE_SIZE")
if [ "${L2:-0}" = "1" ]; then PASSES+=("TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"); fi
for c in "${PASSES[@]}"; do
  tag=$(echo $c | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$OUT/raw -o p -- python $R/bench.py --steps 3 --warmup 1 --inflight 1 --rotate 1 --no-extra-legs --no-graph --no-cpu-baseline --no-other-workloads --no-collective-leg --no-parity-leg --min-seconds 0 --prime 0 "$@" > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/$OUT/raw -name "p_results.db" | head -1) --by-grid > $R/gpurun_out/$OUT/${tag}.txt 2>&1
  rm -rf $R/gpurun_out/$OUT/raw
done
if [ "${STATS:-1}" = "1" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$OUT/stats -o s -- python $R/bench.py --steps 100 --no-cpu-baseline --no-extra-legs --no-other-workloads --no-collective-leg "$@" > $R/gpurun_out/$OUT/bench_under_rocprof.json 2>/dev/null
  python $R/tools/rocpd_stats.py $(find $R/gpurun_out/$OUT/stats -name "s_results.db" | head -1) > $R/gpurun_out/$OUT/kernel_stats.txt 2>&1
  rm -rf $R/gpurun_out/$OUT/stats
  head -30 $R/gpurun_out/$OUT/kernel_stats.txt
fi
