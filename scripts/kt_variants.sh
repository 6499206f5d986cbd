# per-variant kernel averages of the one-stream frame (temporary env-selected variants of one kernel)
# usage: VAR=LIDF_RR_VARIANT VALS="0 1 2" PAT=ray_reduce bash scripts/kt_variants.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ktv; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in $VALS; do
  export $VAR=$v
  rm -rf /tmp/p_v; rocprofv3 --kernel-trace --stats -d /tmp/p_v -o r -- python $R/bench.py --workload e2e --e2e-mode frame --no-side-stream --steps 30 --warmup 3 --no-rocprof > $O/bench_$v.json 2>/dev/null
  python - "$v" "$PAT" <<'PY'
import sqlite3, sys, json, os
cur = sqlite3.connect('/tmp/p_v/r_results.db').cursor()
pat = sys.argv[2].split(",")
rows = list(cur.execute("select name,count(*),avg(duration),min(duration) from kernels group by name"))
tot = sum(r[1] * r[2] for r in rows) / 33
for r in rows:
    if any(p in r[0] for p in pat):
        print("variant %s  %-60s n %4d avg %8.1f us  min %8.1f us" % (sys.argv[1], r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3))
try:
    b = json.loads(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "ktv", "bench_%s.json" % sys.argv[1])).read().strip().splitlines()[-1])
    print("variant %s  busy/step %.1f us  ms_per_frame (under the profiler) %s" % (sys.argv[1], tot / 1e3, b.get("ms_per_frame")))
except Exception as e:
    print("bench record:", e)
PY
done
