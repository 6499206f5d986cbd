#!/bin/bash
# A/B of two builds of the library on the headline query in ONE gpurun call: the tree's liblidf_hip.so against $1;
# prints ms per step and the per-point kernel's HIP-event time, alternating, 3 repetitions
R=$GRAFT_REPO_ROOT; ALT=$1; shift
for rep in 1 2 3; do for v in new old; do
  if [ $v = new ]; then unset LIDF_HIP_LIB; else export LIDF_HIP_LIB=$R/$ALT; fi
  python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-rocprof --no-split-f16 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('headline $v', r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['frac'])"
done; done
unset LIDF_HIP_LIB
