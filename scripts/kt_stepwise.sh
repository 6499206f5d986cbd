cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_l; rocprofv3 --kernel-trace --stats -d /tmp/p_l -o r -- python $GRAFT_REPO_ROOT/bench.py --workload e2e --e2e-mode stepwise --steps 6 --warmup 2 --no-rocprof > /dev/null 2>&1; python - <<'PY'
import sqlite3
cur=sqlite3.connect('/tmp/p_l/r_results.db').cursor()
rows = list(cur.execute("select name,count(*),avg(duration),sum(duration) from kernels group by name order by sum(duration) desc"))
steps = 8
for r in rows[:45]: print("  %-70s x%5.1f  %8.1f us  %8.1f us/frame" % (r[0][:70], r[1]/steps, r[2]/1e3, r[3]/steps/1e3))
print("  total %.1f us/frame, %d launches/frame" % (sum(r[3] for r in rows)/steps/1e3, sum(r[1] for r in rows)/steps))
PY
