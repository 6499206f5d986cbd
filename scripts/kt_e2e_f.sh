# per-kernel time of the evaluation path at $1 frames per call, one stream (no overlap: durations add up)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_ef; rocprofv3 --kernel-trace --stats -d /tmp/p_ef -o r -- python $R/bench.py --workload e2e --e2e-mode frame --frames $1 --no-side-stream --steps 12 --warmup 3 --no-rocprof > /tmp/ef.json 2>/dev/null
python - "$1" <<'PY'
import sqlite3, sys, json
F=int(sys.argv[1])
try: print("ms_per_frame", json.loads(open('/tmp/ef.json').read().strip().splitlines()[-1])["ms_per_frame"])
except Exception as e: print(e)
cur=sqlite3.connect('/tmp/p_ef/r_results.db').cursor()
rows=list(cur.execute("select name,start,end from kernels order by start"))
marks=[i for i,r in enumerate(rows) if "lidf_points_fused_kernel" in r[0]]
a,b=marks[3],marks[-1]; n=len(marks)-4
per={}
for r in rows[a:b]: per.setdefault(r[0].split("(")[0][:52],[]).append(r[2]-r[1])
tot=sum(sum(v) for v in per.values())
print("steps %d, busy per step %.3f ms, per frame %.3f; wall per step %.3f" % (n, tot/n/1e6, tot/n/1e6/F, (rows[b][1]-rows[a][1])/n/1e6))
for k,v in sorted(per.items(), key=lambda kv:-sum(kv[1]))[:16]: print("  %-52s x%5.1f  %8.1f us each  %7.3f ms/step  %6.1f us/frame" % (k, len(v)/n, sum(v)/len(v)/1e3, sum(v)/n/1e6, sum(v)/n/1e3/F))
PY
