O=gpurun_out/s2; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_qr; rocprofv3 --kernel-trace --stats -d /tmp/p_qr -o r -- python $R/bench.py --workload query+refine --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/qr_prof.json 2>/dev/null
cd $R
python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_qr/r_results.db').cursor()
n=13
rows=list(cur.execute("select name,count(*),sum(duration),avg(duration) from kernels group by name order by sum(duration) desc"))
print("busy per step %.3f ms, launches/step %.1f"%(sum(r[2] for r in rows)/n/1e6, sum(r[1] for r in rows)/n))
for r in rows[:30]: print("%-62s %5.1f/step %8.1f us/step avg %7.1f"%(r[0][:62],r[1]/n,r[2]/n/1e3,r[3]/1e3))
PY
