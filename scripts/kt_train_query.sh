O=gpurun_out/s2; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tq; rocprofv3 --kernel-trace --stats -d /tmp/p_tq -o r -- python $R/bench.py --workload train-query --steps 10 --warmup 3 > $R/$O/tq_prof.json 2>/dev/null
cd $R
python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_tq/r_results.db').cursor()
n=13
rows=list(cur.execute("select name,count(*),sum(duration),avg(duration) from kernels group by name order by sum(duration) desc"))
print("busy per step %.3f ms, launches/step %.1f"%(sum(r[2] for r in rows)/n/1e6, sum(r[1] for r in rows)/n))
for r in rows[:45]: print("%-62s %5.1f/step %8.1f us/step avg %7.1f"%(r[0][:62],r[1]/n,r[2]/n/1e3,r[3]/1e3))
PY
cut -c1-300 $O/tq_prof.json
