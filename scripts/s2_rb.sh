R=$GRAFT_REPO_ROOT
for lib in "" scripts/liblidf_rb6.so scripts/liblidf_rb8.so; do
  if [ -n "$lib" ]; then export LIDF_HIP_LIB=$R/$lib; else unset LIDF_HIP_LIB; fi
  echo "== lib: ${lib:-default (RB=4)}"
  for p in dense scene n1; do
    python bench.py --pairs $p --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  $p', d['value'], 'Mp/s step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'])"
  done
  python bench.py --workload e2e --frames 4 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  e2e x4', d['ms_per_step'], d['stage_ms']['query'])"
done
