# effective shader clock and matrix-pipe occupancy of the training step's kernels:
# clock = SQ_BUSY_CYCLES / 32 (XCD x SE) / duration, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (clock x duration)
O=gpurun_out/s2; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_tp; rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/p_tp -o r -- python $R/bench.py --workload ${1:-train-query} --steps 4 --warmup 2 > /dev/null 2>&1
cd $R
python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_tp/r_results.db').cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
rows=list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
dur={r[0]:(r[1],r[2]) for r in cur.execute("select name,count(*),avg(duration) from kernels group by name")}
by={}
for k,c,n,v in rows: by.setdefault(k,{})[c]=v
print("%-64s %9s %8s %8s"%("kernel","us","GHz","mfma"))
for k,d in sorted(by.items(), key=lambda kv:-dur.get(kv[0],(0,0))[1]*dur.get(kv[0],(0,0))[0])[:14]:
    if k not in dur or 'SQ_BUSY_CYCLES' not in d: continue
    t=dur[k][1]*1e-9; clk=d['SQ_BUSY_CYCLES']/32/t
    print("%-64s %9.1f %8.3f %8.3f"%(k[:64], t*1e6, clk/1e9, d.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/(clk*t)))
PY
