# kernel traces of the evaluation path pipelined over 3 streams, this tree and _old/, in one session
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in new old new old; do
  D=$R; [ $v = old ] && D=$R/_old
  rm -rf /tmp/k3_$v
  (cd $D && rocprofv3 --kernel-trace --stats -d /tmp/k3_$v -o r -- python bench.py --workload e2e --e2e-mode frame --streams 3 --steps 200 --warmup 20 --no-rocprof > /dev/null 2>&1)
  python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/k3_$v/r_results.db').cursor()
rows=list(cur.execute("select name,start,end,duration,stream_id from kernels order by start"))
# steady state: from the 40th to the 200th launch of the per-point kernel
pts=[r for r in rows if 'lidf_points_fused' in r[0]]
t0,t1=pts[40][1],pts[200][1]
sel=[r for r in rows if t0<=r[1]<t1]
busy=sum(r[3] for r in sel)
# union of busy intervals
iv=sorted((r[1],r[2]) for r in sel); cov=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: cov+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
cov+=ce-cs
print("$v frames 160 span_ms %.2f per_frame %.4f busy_sum_ms %.2f covered_ms %.2f idle_ms %.2f"%((t1-t0)/1e6,(t1-t0)/1e6/160,busy/1e6,cov/1e6,(t1-t0-cov)/1e6))
import collections
d=collections.defaultdict(list)
for r in sel: d[r[0].split('(')[0][:44]].append(r[3])
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:8]: print("    %5d x %8.1f us %s"%(len(v), sum(v)/len(v)/1e3, k))
PY
done
