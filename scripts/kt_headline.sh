# kernel-trace of the headline step: per-kernel average durations -> gpurun_out/s2/kt_<tag>.txt
TAG=${1:-x}; O=gpurun_out/s2; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt; rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/bench_$TAG.json 2>/dev/null
cd $R
python - <<PY > $O/kt_$TAG.txt
import sqlite3
cur=sqlite3.connect('/tmp/p_kt/r_results.db').cursor()
for r in cur.execute("select name,count(*),avg(duration),min(duration) from kernels group by name order by sum(duration) desc limit 14"):
    print("%-50s %4d avg %9.1f us  min %9.1f"%(r[0][:50],r[1],r[2]/1e3,r[3]/1e3))
PY
cat $O/kt_$TAG.txt; python -c "
import json; d=json.load(open('$O/bench_$TAG.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_kt/r_results.db').cursor()
for r in cur.execute("select name,max(vgpr_count),max(accum_vgpr_count),max(sgpr_count),max(scratch_size) from kernels where name like '%points_fused%' group by name"): print(r)
PY
