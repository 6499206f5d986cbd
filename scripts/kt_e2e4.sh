O=gpurun_out/s2; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_e4; rocprofv3 --kernel-trace --stats -d /tmp/p_e4 -o r -- python $R/bench.py --workload e2e --frames 4 --steps 10 --warmup 3 > $R/$O/e2e4_prof.json 2>/dev/null
cd $R
python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_e4/r_results.db').cursor()
rows=list(cur.execute("select name,count(*),sum(duration),avg(duration) from kernels group by name order by sum(duration) desc limit 32"))
n=13
print("busy per step %.3f ms"%(sum(r[2] for r in cur.execute("select name,count(*),sum(duration) from kernels group by name"))/n/1e6))
for r in rows: print("%-60s %5.1f/step %9.1f us/step avg %8.1f"%(r[0][:60],r[1]/n,r[2]/n/1e3,r[3]/1e3))
PY
cut -c1-900 $O/e2e4_prof.json
