# where the waves of the frame's matrix kernels wait: per-kernel SQ counters (separate passes), e2e frame mode
O=gpurun_out/pmcw; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
i=$((i+1)); rm -rf /tmp/p_w$i
rocprofv3 --kernel-trace --pmc $C -d /tmp/p_w$i -o r -- python $R/bench.py --workload e2e --e2e-mode frame --no-side-stream --steps 3 --warmup 2 --no-rocprof > /dev/null 2> /tmp/p_w$i.err || tail -2 /tmp/p_w$i.err
done
cd $R
python - > $O/waits.txt <<'PY'
import sqlite3, glob
by = {}
dur = {}
for db in sorted(glob.glob('/tmp/p_w*/r_results.db')):
    cur = sqlite3.connect(db).cursor()
    for k, c, v in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        by.setdefault(k, {})[c] = v
    for k, n, d in cur.execute("select name, count(*), avg(duration) from kernels group by name"):
        dur[k] = d
for k in sorted(by, key=lambda k: -dur.get(k, 0))[:8]:
    print("==", k[:70], "%.1f us" % (dur.get(k, 0) / 1e3))
    for c, v in sorted(by[k].items()):
        print("   %-34s %.6g" % (c, v))
PY
cat $O/waits.txt | head -150
