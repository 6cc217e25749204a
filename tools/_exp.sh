timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --no-header -p no:cacheprovider -W ignore -x 2>&1 | tail -2
for bk in 64 128; do
MV2D_BF16_BK=$bk timeout 120 python bench.py --steps 100 --inflight 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bk $bk', d['value'], {k:round(v*1000,1) for k,v in d['stage_ms'].items() if 'gemm' in k})"
MV2D_BF16_BK=$bk timeout 120 python bench.py --steps 100 --inflight 1 --no-cpu-baseline --workload cfg5_t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bk $bk cfg5_t', d['value'], {k:round(v*1000,1) for k,v in d['stage_ms'].items() if 'gemm' in k})"
done
