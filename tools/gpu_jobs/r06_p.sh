for lib in base mv2d_amd/lib/variants/libpxr5.so; do
  if [ "$lib" = base ]; then unset MV2D_HIP_LIB; else export MV2D_HIP_LIB=$lib; fi
  python tools/pe_time.py 250000 1 2>&1 | grep "pe_x3 "; python tools/pe_time.py 100000 0 2>&1 | grep "pe_x3 "
  for a in "cfg2_s 16" "cfg3_t 16"; do set -- $a; python bench.py --workload $1 --batch $2 --steps 40 --warmup 10 --brief --no-parity-leg 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(\"[$lib] $a\", d[\"value\"], \"pe_fused stage ms\", d[\"stage_ms\"][\"pe_fused\"])"; done
done
