cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/pt -o p -- python /root/repo/bench.py --workload ${1:-train-query} --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("/tmp/pt/**/*.db",recursive=True)[0]
rows=list(sqlite3.connect(db).execute("select name,start,duration,grid_x from kernels order by start"))
n=len(rows)//3
t0=rows[-n][1]
for r in rows[-n:]:
    print("%8.1f %7.1f %s g=%d" % ((r[1]-t0)/1e3, r[2]/1e3, r[0][:60], r[3]))
PY
