# per-kernel averages of the evaluation path at $1 frames per call (one stream) — the kernels without their round quantisation
F=${1:-16}
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p_l; rocprofv3 --kernel-trace --stats -d /tmp/p_l -o r -- python $GRAFT_REPO_ROOT/bench.py --workload e2e --e2e-mode frame --no-side-stream --frames $F --steps 6 --warmup 2 --no-rocprof > /dev/null 2>&1; python - $F <<'PY'
import sqlite3, sys
F = int(sys.argv[1])
cur=sqlite3.connect('/tmp/p_l/r_results.db').cursor()
rows = list(cur.execute("select name,count(*),avg(duration),sum(duration) from kernels group by name order by sum(duration) desc"))
steps = 8
for r in rows[:22]: print("  %-64s x%5.1f /call  %8.1f us  %8.1f us/frame" % (r[0][:64], r[1]/steps, r[2]/1e3, r[3]/steps/F/1e3))
print("  total %.1f us/frame" % (sum(r[3] for r in rows)/steps/F/1e3))
PY
