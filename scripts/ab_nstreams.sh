# frames pipelined over S streams (one FrameRunner per stream): S = 3..12
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/abs; mkdir -p $O
cd $R
for rep in 1 2 3; do for s in 3 5 6 8 12; do
  python bench.py --workload e2e --e2e-mode frame --streams $s --steps 300 --warmup 30 --no-rocprof > $O/s${s}_$rep.json 2>/dev/null
done; done
for rep in 1 2; do for s in 2 3 4; do
  python bench.py --workload e2e --e2e-mode frame --streams $s --frames 4 --steps 80 --warmup 8 --no-rocprof > $O/f4_s${s}_$rep.json 2>/dev/null
  python bench.py --workload e2e --e2e-mode frame --streams $((2*s)) --offsets selected --steps 300 --warmup 30 --no-rocprof > $O/sel_s$((2*s))_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "abs")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-16s ms/frame %s" % (os.path.basename(f)[:-5], r.get("ms_per_frame")))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
