#!/bin/bash
# Frames pipelined over 3 streams, this tree against _old/ (git archive of the round-4 tree, built in place), ONE gpurun
# session, alternating, 2,000 frames after 300 — and the same with 8 hardware queues. Then the instruction-cache
# counters of the frame's kernels in both trees (one stream; counter runs serialise dispatches).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_s3; mkdir -p $O
N=${1:-2000}; W=${2:-300}
for rep in 1 2 3; do for v in new old; do
  D=$R; [ $v = old ] && D=$R/_old
  for q in def 8; do
    unset GPU_MAX_HW_QUEUES; [ $q != def ] && export GPU_MAX_HW_QUEUES=$q
    (cd $D && python bench.py --workload e2e --e2e-mode frame --streams 3 --steps $N --warmup $W --no-rocprof > $O/s3_${v}_q${q}_$rep.json 2>/dev/null)
  done
  unset GPU_MAX_HW_QUEUES
  (cd $D && python bench.py --workload e2e --e2e-mode frame --steps $N --warmup $W --no-rocprof > $O/s1_${v}_$rep.json 2>/dev/null)
done; done
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "ab_s3")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-18s ms/frame %.4f" % (os.path.basename(f)[:-5], r["ms_per_frame"]))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  D=$R; [ $v = old ] && D=$R/_old
  rm -rf /tmp/p_ic_$v
  (cd $D && rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_BUSY_CYCLES -d /tmp/p_ic_$v -o r -- python bench.py --workload e2e --e2e-mode frame --no-side-stream --steps 6 --warmup 3 --no-rocprof > /dev/null 2>&1)
  python - $v <<'PY'
import sqlite3, sys, glob
v = sys.argv[1]
dbs = glob.glob("/tmp/p_ic_%s/**/*_results.db" % v, recursive=True)
if not dbs:
    print(v, "no counter db"); sys.exit(0)
cur = sqlite3.connect(dbs[0]).cursor()
dur = {r[0]: (r[1], r[2]) for r in cur.execute("select name, count(*), avg(duration) from kernels group by name")}
by = {}
for k, c, a in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
    by.setdefault(k, {})[c] = a
print("tree %s: instruction cache per launch (rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE)" % v)
print("%-60s %9s %12s %12s %10s %8s" % ("kernel", "us", "req", "misses", "dup", "miss %"))
for k, d in sorted(by.items(), key=lambda kv: -dur.get(kv[0], (0, 0))[1] * dur.get(kv[0], (0, 0))[0])[:9]:
    if k not in dur: continue
    rq, ms = d.get("SQC_ICACHE_REQ", 0), d.get("SQC_ICACHE_MISSES", 0)
    print("%-60s %9.1f %12.0f %12.0f %10.0f %8.3f" % (k.split("(")[0][:60], dur[k][1] / 1e3, rq, ms, d.get("SQC_ICACHE_MISSES_DUPLICATE", 0), 100.0 * ms / max(rq, 1)))
PY
done
