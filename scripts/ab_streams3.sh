R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab3; mkdir -p $O
for rep in 1 2 3; do for v in split whole old; do
  D=$R; [ $v = old ] && D=$R/_old
  unset LIDF_TAIL_SPLIT; [ $v = whole ] && export LIDF_TAIL_SPLIT=0
  (cd $D && python bench.py --workload e2e --e2e-mode frame --streams 3 --steps 300 --warmup 30 --no-rocprof > $O/s3_${v}_$rep.json 2>/dev/null)
  (cd $D && python bench.py --workload e2e --e2e-mode frame --streams 2 --frames 4 --steps 80 --warmup 8 --no-rocprof > $O/s2f4_${v}_$rep.json 2>/dev/null)
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "ab3")
for f in sorted(glob.glob(O + "/*.json")):
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print("%-16s ms/frame %s" % (os.path.basename(f)[:-5], r.get("ms_per_frame")))
PY
