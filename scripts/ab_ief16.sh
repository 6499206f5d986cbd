# A/B inside one GPU session (box-to-box variance is 2-3 %): the stage-2 decoder on 16 x 16 x 4 sub-tiles
# (default) against the 32 x 32 rows kernel of rounds 2-3 (LIDF_IEF16=0), same frame, alternating
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in new old new old; do
  if [ $v = new ]; then unset LIDF_IEF16; else export LIDF_IEF16=0; fi
  rm -rf /tmp/p_$v; rocprofv3 --kernel-trace --stats -d /tmp/p_$v -o r -- python $R/bench.py --workload e2e --e2e-mode frame --steps 20 --warmup 3 --no-rocprof > /dev/null 2>&1
  python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_$v/r_results.db').cursor()
print("$v", " | ".join("%s x%d %.1f us"%(r[0][:34], r[1], r[2]/1e3) for r in cur.execute("select name,count(*),avg(duration) from kernels where name like '%ief16%' or name like '%points_kernel<6>%' group by name")))
PY
done
