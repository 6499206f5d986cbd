R=$GRAFT_REPO_ROOT
for lib in "" scripts/liblidf_c2.so scripts/liblidf_c1.so; do
  if [ -n "$lib" ]; then export LIDF_HIP_LIB=$R/$lib; else unset LIDF_HIP_LIB; fi
  echo "== lib: ${lib:-default (chunk 4)}"
  for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  dense', d['value'], 'step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
  python bench.py --pairs ragged --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  ragged', d['value'], 'step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
  python bench.py --workload e2e --frames 4 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  e2e x4', d['ms_per_step'], d['stage_ms']['query'])"
done
