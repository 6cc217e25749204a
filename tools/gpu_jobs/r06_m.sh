MV2D_HIP_LIB=mv2d_amd/lib/variants/libxftrace.so python tools/xf_trace.py cfg2_s 2>&1 | grep -v Warn | tail -16
MV2D_HIP_LIB=mv2d_amd/lib/variants/libxftrace.so python tools/xf_trace.py cfg2_s_nc6 2>&1 | grep -v Warn | tail -16
for qb in 8 4; do MV2D_XF_QB=$qb python bench.py --steps 40 --warmup 10 --brief --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('QB=$qb', d['value'], d['roofline'].get('launch_ms_idle_gpu'))"; done
