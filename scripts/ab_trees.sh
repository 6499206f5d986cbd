#!/bin/bash
# The tree against a second tree under _old/ (a `git archive` of an earlier commit, built in place) in ONE gpurun
# session, alternating: whole-program A/B (library + Python) of the evaluation path. usage: ab_trees.sh "<bench args>" ...
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_trees; mkdir -p $O
i=0
for ARGS in "$@"; do
  i=$((i+1))
  for rep in 1 2; do for v in new old; do
    D=$R; [ $v = old ] && D=$R/_old
    (cd $D && python bench.py $ARGS --no-rocprof > $O/a${i}_${v}_$rep.json 2>/dev/null)
  done; done
done
python - "$@" <<'PY'
import json, glob, os, sys
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "ab_trees")
for i, a in enumerate(sys.argv[1:], 1):
    print("bench.py", a)
    for f in sorted(glob.glob(O + "/a%d_*.json" % i)):
        try:
            r = json.loads(open(f).read().strip().splitlines()[-1])
            print("   %-16s ms/step %.4f  ms/frame %s" % (os.path.basename(f)[:-5], r["ms_per_step"], r.get("ms_per_frame")))
        except Exception as e:
            print("   ", os.path.basename(f), "ERR", e)
PY
