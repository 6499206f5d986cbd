# the query at another decoder width (bench.py --imnet-gf): the chain launch (default) against the layer-by-layer path
# (LIDF_CHAIN16=0), one session, alternating
R=$GRAFT_REPO_ROOT
for gf in 32 128; do for v in chain layers chain layers; do
  if [ $v = chain ]; then unset LIDF_CHAIN16; else export LIDF_CHAIN16=0; fi
  python $R/bench.py --imnet-gf $gf --steps 3 --warmup 1 --no-rocprof --no-cpu-baseline --no-split-f16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gf $gf $v', d['value'], d['unit'], d['ms_per_step'], 'ms frac', d['roofline'].get('frac'), 'parity', (d.get('parity') or {}).get('ok'))"
done; done
