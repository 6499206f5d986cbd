# A/B of two builds of the library in ONE gpurun call (box-to-box variance is 2-3 %): the tree's
# liblidf_hip.so against $1 (a second build kept at the repo root), kernels matching $2, workload $3
R=$GRAFT_REPO_ROOT; ALT=$1; PAT=${2:-lidf}; W=${3:-train-query}
cd /tmp && export TMPDIR=/tmp
for v in new old new old; do
  if [ $v = new ]; then unset LIDF_HIP_LIB; else export LIDF_HIP_LIB=$R/$ALT; fi
  rm -rf /tmp/p_$v; rocprofv3 --kernel-trace --stats -d /tmp/p_$v -o r -- python $R/bench.py --workload $W --steps 10 --warmup 3 > /dev/null 2>&1
  python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_$v/r_results.db').cursor()
n=13
print("$v", " | ".join("%s x%.0f %.1f"%(r[0][:34], r[1]/n, r[2]/1e3) for r in cur.execute("select name,count(*),avg(duration) from kernels where name like '%$PAT%' group by name")))
PY
done
