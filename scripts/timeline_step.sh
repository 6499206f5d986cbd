# one steady-state step of a training workload as a timeline: kernel, queue, start offset, duration, gap to the previous
# launch on the same queue. usage: timeline_step.sh <workload> <marker kernel substring> [bench flags]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tl; rocprofv3 --kernel-trace --stats -d /tmp/p_tl -o r -- python $R/bench.py --workload $1 --steps 6 --warmup 3 --no-rocprof "${@:3}" > /dev/null 2>&1
python - "$2" <<'PY'
import sqlite3, sys
cur=sqlite3.connect('/tmp/p_tl/r_results.db').cursor()
cols=[r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows=list(cur.execute("select name,start,end,%s from kernels order by start" % (qcol or "0")))
marks=[i for i,r in enumerate(rows) if sys.argv[1] in r[0]]
a,b=marks[-3],marks[-2]
t0=rows[a][1]; last={}
print("queue column:", qcol, "| step wall %.3f ms" % ((rows[b][1]-t0)/1e6))
busy={}
for r in rows[a:b]:
    q=r[3]; gap=(r[1]-last[q])/1e3 if q in last else 0.0; last[q]=r[2]
    busy[q]=busy.get(q,0)+(r[2]-r[1])
    print("q%-3s %9.1f us  +%8.1f us  gap %7.1f  %s" % (q, (r[1]-t0)/1e3, (r[2]-r[1])/1e3, gap, r[0][:70]))
print({q: round(v/1e6,3) for q,v in busy.items()})
PY
