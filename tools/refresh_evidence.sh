# Final evidence of a round: default bench line (with the CPU baseline), the two T-path workloads, the one-sample-per-call shapes.
# usage (on the GPU box): bash tools/refresh_evidence.sh <outdir-under-gpurun_out>
OUT=gpurun_out/${1:-evidence}
mkdir -p $OUT
timeout 400 python bench.py --steps 100 > $OUT/bench_default.json 2> $OUT/bench.err
timeout 300 python bench.py --steps 100 --workload cfg3_t --no-cpu-baseline > $OUT/bench_cfg3_t.json 2>> $OUT/bench.err
timeout 300 python bench.py --steps 100 --workload cfg5_t --batch 2 --no-cpu-baseline > $OUT/bench_cfg5_t.json 2>> $OUT/bench.err
tail -c 300 $OUT/bench.err
