# frames pipelined over 3 streams: sensitivity to the number of hardware queues HIP maps its streams onto
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abq; mkdir -p $O
cd $R
for rep in 1 2 3; do for q in def 2 3 4 8; do
  unset GPU_MAX_HW_QUEUES; [ $q != def ] && export GPU_MAX_HW_QUEUES=$q
  python bench.py --workload e2e --e2e-mode frame --streams 3 --steps 300 --warmup 30 --no-rocprof > $O/s3_q${q}_$rep.json 2>/dev/null
done; done
unset GPU_MAX_HW_QUEUES
for rep in 1 2 3; do for s in 2 4 6; do
  python bench.py --workload e2e --e2e-mode frame --streams $s --steps 300 --warmup 30 --no-rocprof > $O/s${s}_qdef_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "abq")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-16s ms/frame %s" % (os.path.basename(f)[:-5], r.get("ms_per_frame")))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
