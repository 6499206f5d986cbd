# kernel trace of one training step of the query (bench.py --workload train-query): per-kernel totals
# and the launches of the last step in time order
O=gpurun_out/s2; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tq; rocprofv3 --kernel-trace --stats -d /tmp/p_tq -o r -- python $R/bench.py --workload train-query --steps 10 --warmup 3 --no-rocprof > $R/$O/tq_prof.json 2>/dev/null
cd $R
python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_tq/r_results.db').cursor()
n=13
rows=list(cur.execute("select name,count(*),sum(duration),avg(duration) from kernels group by name order by sum(duration) desc"))
print("busy per step %.3f ms, launches/step %.1f"%(sum(r[2] for r in rows)/n/1e6, sum(r[1] for r in rows)/n))
for r in rows[:60]: print("%-62s %5.1f/step %8.1f us/step avg %7.1f"%(r[0][:62],r[1]/n,r[2]/n/1e3,r[3]/1e3))
ks=list(cur.execute("select name,start,end,grid_x,workgroup_x from kernels order by start"))
per=len(ks)//n
last=ks[-per:]
t0=last[0][1]; prev=None
print("--- last step in time order: start us, duration us, gap us, grid")
for nm,s,e,g,wg in last:
    print("%9.1f %8.1f %7.1f %8d  %s"%((s-t0)/1e3,(e-s)/1e3,0 if prev is None else (s-prev)/1e3,g//max(wg,1),nm[:70])); prev=e
PY
cut -c1-200 $O/tq_prof.json
