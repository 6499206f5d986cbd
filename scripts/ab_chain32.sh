# gf 32 chain launch: wavefronts per SIMD x sub-tiles side by side (LIDF_CHAIN16_GF32), one session, alternating
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for v in 22 21 41; do
  export LIDF_CHAIN16_GF32=$v
  python $R/bench.py --imnet-gf 32 --steps 3 --warmup 1 --no-rocprof --no-cpu-baseline --no-split-f16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gf 32 cfg $v', d['value'], d['unit'], d['ms_per_step'], 'ms frac', d['roofline'].get('frac'))"
done; done
