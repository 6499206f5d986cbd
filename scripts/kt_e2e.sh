# kernel trace of the whole evaluation path (bench.py --workload e2e): busy time per step against
# the wall clock of a step, and the launches of one step in time order with the gaps between them
F=${1:-1}; M=${2:-frame}; O=gpurun_out/e2e_kt${F}_$M; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_e; rocprofv3 --kernel-trace --stats -d /tmp/p_e -o r -- python $R/bench.py --workload e2e --frames $F --e2e-mode $M --steps 10 --warmup 3 $KT_OPTS > $R/$O/bench.json 2>/dev/null
cd $R
python - > $O/summary.txt <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_e/r_results.db').cursor()
n=13
rows=list(cur.execute("select name,count(*),sum(duration),avg(duration) from kernels group by name order by sum(duration) desc"))
print("busy per step %.3f ms, launches per step %.1f"%(sum(r[2] for r in rows)/n/1e6, sum(r[1] for r in rows)/n))
for r in rows[:40]: print("%-70s %5.1f/step %9.1f us/step avg %8.1f"%(r[0][:70],r[1]/n,r[2]/n/1e3,r[3]/1e3))
ks=list(cur.execute("select name,start,end from kernels order by start"))
per=len(ks)//n
last=ks[-per:]
print("--- last step, in time order: start offset us, duration us, gap before us")
t0=last[0][1]; prev=None
for nm,s,e in last:
    print("%9.1f %8.1f %8.1f  %s"%((s-t0)/1e3,(e-s)/1e3,0 if prev is None else (s-prev)/1e3,nm[:80])); prev=e
print("step span %.3f ms"%((last[-1][2]-t0)/1e6))
PY
cut -c1-1500 $O/bench.json; head -50 $O/summary.txt
