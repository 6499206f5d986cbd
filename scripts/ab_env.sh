#!/bin/bash
# A/B of an environment knob in ONE gpurun call: usage ab_env.sh VAR "v1 v2 ..." workload [workload ...]; "-" = unset
R=$GRAFT_REPO_ROOT; VAR=$1; VALS=$2; shift 2
for W in "$@"; do for rep in 1 2; do for v in $VALS; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
  python $R/bench.py --workload $W --steps 20 --warmup 5 --no-rocprof 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$W', '$VAR=$v', r['ms_per_step'])"
done; done; done
