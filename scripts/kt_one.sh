# average duration of the kernels matching $1 in one e2e frame run (one stream), under rocprofv3
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_l; rocprofv3 --kernel-trace --stats -d /tmp/p_l -o r -- python $GRAFT_REPO_ROOT/bench.py --workload e2e --e2e-mode frame --no-side-stream --steps 6 --warmup 2 --no-rocprof ${@:2} > /dev/null 2>&1; python - "$1" <<'PY'
import sqlite3, sys
cur=sqlite3.connect('/tmp/p_l/r_results.db').cursor()
for r in cur.execute("select name,count(*),avg(duration) from kernels where name like ? group by name", ('%'+sys.argv[1]+'%',)): print("  ", r[0][:60], r[1], "%.1f us"%(r[2]/1e3))
PY
