cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06tl
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/r06tl/st -o s -- python $R/bench.py --steps 30 --warmup 5 --brief --no-parity-leg --no-cpu-baseline --no-other-workloads --no-extra-legs --no-collective-leg > /dev/null 2>&1
DB=$(find $R/gpurun_out/r06tl/st -name "s_results.db" | head -1)
python $R/tools/rocpd_concurrency.py $DB 200 | tee $R/gpurun_out/r06tl/concurrency.txt
rm -rf $R/gpurun_out/r06tl/st
