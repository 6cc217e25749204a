# HBM-side counters of the eager single-stream bench (separate --pmc passes, kernel trace only) + kernel-trace stats of the default bench
# usage: tools/pmc_bench.sh [outdir-under-gpurun_out]   (default pmc2)
OUT=${1:-pmc2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$OUT/$c -o p -- python $R/bench.py --steps 3 --warmup 1 --inflight 1 --rotate 1 --no-extra-legs --no-graph --no-cpu-baseline --no-other-workloads --no-parity-leg --min-seconds 0 --prime 0 > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $(find $R/gpurun_out/$OUT/$c -name "p_results.db" | head -1) --by-grid > $R/gpurun_out/$OUT/${c}.txt 2>&1
  rm -rf $R/gpurun_out/$OUT/$c
done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$OUT/stats -o s -- python $R/bench.py --steps 100 --no-cpu-baseline --no-extra-legs --no-other-workloads > $R/gpurun_out/$OUT/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocpd_stats.py $(find $R/gpurun_out/$OUT/stats -name "s_results.db" | head -1) > $R/gpurun_out/$OUT/kernel_stats.txt 2>&1
rm -rf $R/gpurun_out/$OUT/stats
head -30 $R/gpurun_out/$OUT/kernel_stats.txt
