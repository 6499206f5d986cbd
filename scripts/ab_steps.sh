# A/B of two builds of the library in ONE gpurun call on the step time of some workloads (box-to-box variance is
# 2-3 %): the tree's liblidf_hip.so against $1 (a second build kept at the repo root); workloads = the remaining arguments
R=$GRAFT_REPO_ROOT; ALT=$1; shift
for W in "$@"; do
  for v in new old new old; do
    if [ $v = new ]; then unset LIDF_HIP_LIB; else export LIDF_HIP_LIB=$R/$ALT; fi
    python $R/bench.py --workload $W --steps 20 --warmup 5 --no-rocprof 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$W', '$v', r['ms_per_step'])"
  done
done
