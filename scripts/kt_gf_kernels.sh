# per-kernel time of the query at another decoder width ($1 = gf), one step of 4,915,200 pairs
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_gf; rocprofv3 --kernel-trace --stats -d /tmp/p_gf -o r -- python $R/bench.py --imnet-gf $1 --steps 3 --warmup 1 --no-rocprof --no-cpu-baseline --no-split-f16 > /dev/null 2>&1
python - <<'PY'
import sqlite3
cur=sqlite3.connect('/tmp/p_gf/r_results.db').cursor()
rows=list(cur.execute("select name,count(*),sum(duration),avg(duration) from kernels group by name order by sum(duration) desc"))
tot=sum(r[2] for r in rows)
for r in rows[:14]: print("%-70s x%-5d total %8.2f ms  avg %8.1f us  %4.1f%%"%(r[0][:70], r[1], r[2]/1e6, r[3]/1e3, 100*r[2]/tot))
PY
