"""Round-6 profile summaries: gpurun_out/r6p (scripts/r06_runs.sh) -> profiles/r06_*: the bench records as
they were printed, per-kernel statistics of the kernel-trace runs, the HBM and matrix-pipe counters of the
headline, clock / pipe occupancy of the evaluation path's kernels.

Every CSV is taken from a command that runs ONE launch shape per kernel and step — the headline command carries
--no-split-f16 (no split-f16 leg, no offsets="selected" leg, no extra parity step), so `lidf_points_fused_kernel`
has steps + warmup calls of the headline shape and "F x units / avg_ns / peak" from the CSV alone reproduces the
record's fraction (scripts/verify_records.py checks exactly that)."""
import json
import os
import shutil
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r6p")
# run on the GPU box (the rocprofv3 databases are too large to travel back): summaries land in
# gpurun_out/r6prof, which is then copied into profiles/
OUT = os.path.join(ROOT, sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)
TAG = "r06"

HEADLINE = "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rocprof --no-split-f16"
PMC = "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof --no-split-f16"


def stats(db, out, cmd):
    cur = sqlite3.connect(db).cursor()
    per = {}
    meta = {}
    for name, dur, vg, ag, sg, lds, scr, gx, wx in cur.execute(
            "select name, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, "
            "workgroup_x from kernels order by start"):
        per.setdefault(name, []).append(dur)
        m = meta.setdefault(name, [0] * 7)
        for i, v in enumerate((vg, ag, sg, lds, scr, gx, wx)):
            m[i] = max(m[i], v or 0)
    tot = sum(sum(v) for v in per.values()) or 1
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % cmd)
        f.write("# avg_after_first_ns: the average without the kernel's first launch of the process (code load, cold "
                "caches, clock ramp) — what a record's `frac_rocprof` is computed from; avg_ns is over all calls\n")
        f.write("name,calls,total_ns,avg_ns,avg_after_first_ns,min_ns,max_ns,pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,"
                "grid_x,wg_x\n")
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            m = meta[name]
            f.write('"%s",%d,%d,%.1f,%.1f,%d,%d,%.3f,%s,%s,%s,%s,%s,%s,%s\n' % (
                name, len(v), sum(v), sum(v) / len(v), sum(v[1:]) / max(len(v) - 1, 1), min(v), max(v),
                100.0 * sum(v) / tot, *m))


def clock(db, out, cmd):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                            "group by kernel_name, counter_name"))
    dur = {r[0]: (r[1], r[2]) for r in cur.execute("select name, count(*), avg(duration) from kernels group by name")}
    by = {}
    for k, c, v in rows:
        by.setdefault(k, {})[c] = v
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -- %s\n" % cmd)
        f.write("# (the clock of THIS counter run — a profiled child —, not of a timed run)\n")
        f.write("# GHz  = SQ_BUSY_CYCLES / 32 (8 XCDs x 4 shader engines) / kernel duration: the shader clock the launch ran at (spec 2.4)\n")
        f.write("# mfma = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GHz x duration): share of those cycles the matrix pipe was busy\n")
        f.write("%-72s %9s %8s %8s\n" % ("kernel", "us", "GHz", "mfma"))
        for k, d in sorted(by.items(), key=lambda kv: -dur.get(kv[0], (0, 0))[1] * dur.get(kv[0], (0, 0))[0])[:18]:
            if k not in dur or "SQ_BUSY_CYCLES" not in d:
                continue
            t = dur[k][1] * 1e-9
            clk = d["SQ_BUSY_CYCLES"] / 32 / t
            f.write("%-72s %9.1f %8.3f %8.3f\n" % (k[:72], t * 1e6, clk / 1e9,
                                                  d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (clk * t)))


def counter(db, name):
    cur = sqlite3.connect(db).cursor()
    return list(cur.execute(
        "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
        "group by kernel_name order by sum(value) desc", (name,)))


def hbm(fetch_db, write_db, out):
    fetch, write = counter(fetch_db, "FETCH_SIZE"), counter(write_db, "WRITE_SIZE")
    wmap = {r[0]: r for r in write}
    with open(out, "w") as f:
        f.write("# separate passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- %s\n" % PMC)
        f.write("# values are KiB per dispatch as reported; MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE\n")
        f.write("# under-counts wide coalesced streaming reads by 2x (64 B tallied per 128 B request), so\n")
        f.write("# corrected = 2*FETCH_SIZE + WRITE_SIZE is an upper bound for mixed access widths; WRITE_SIZE uncalibrated.\n")
        f.write("kernel,dispatches,fetch_kib_avg,write_kib_avg,hbm_bytes_reported,hbm_bytes_corrected\n")
        for r in fetch:
            w = wmap.get(r[0], (r[0], 0, 0.0))
            f.write('"%s",%d,%.1f,%.1f,%.0f,%.0f\n' % (r[0], r[1], r[2], w[2], (r[2] + w[2]) * 1024,
                                                       (2 * r[2] + w[2]) * 1024))


def mfma(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                            "where kernel_name like '%lidf_points%' group by kernel_name, counter_name"))
    with open(out, "w") as f:
        f.write("# rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -- %s\n" % PMC)
        f.write("# (SQ_INSTS_VALU_MFMA_MOPS_F32 ticks once per 512 f32 MFMA FLOP)\n")
        f.write("kernel,counter,dispatches,avg_value\n")
        for r in rows:
            f.write('"%s",%s,%d,%.6g\n' % (r[0], r[1], r[2], r[3]))


def main():
    j = os.path.join
    for f in sorted(os.listdir(SRC)):
        if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(j(SRC, f)) > 0:
            shutil.copy(j(SRC, f), j(OUT, "%s_%s" % (TAG, f)))
    for f in sorted(os.listdir(SRC)):
        if f.endswith(".txt") and f != "err.txt":
            shutil.copy(j(SRC, f), j(OUT, "%s_%s" % (TAG, f)))
    runs = [("kt.db", "kernel_stats.csv", HEADLINE),
            ("kt_refine.db", "kernel_stats_refine.csv", HEADLINE.replace("bench.py", "bench.py --config 3")),
            ("kt_e2e.db", "kernel_stats_e2e.csv",
             "python bench.py --workload e2e --e2e-mode frame --steps 10 --warmup 3 --no-rocprof"),
            ("kt_e2e_onestream.db", "kernel_stats_e2e_onestream.csv",
             "python bench.py --workload e2e --e2e-mode frame --no-side-stream --steps 10 --warmup 3 --no-rocprof"),
            ("kt_train_refine.db", "kernel_stats_train_refine.csv",
             "python bench.py --workload train-refine --steps 10 --warmup 3 --no-rocprof  (13 steps + 10 host-timing steps)"),
            ("kt_train_query.db", "kernel_stats_train_query.csv",
             "python bench.py --workload train-query --steps 10 --warmup 3 --no-rocprof  (13 steps + 10 host-timing steps)"),
            ("kt_train.db", "kernel_stats_train.csv",
             "python bench.py --workload train --steps 10 --warmup 3 --no-rocprof  (13 steps + 10 host-timing steps)"),
            ("kt_train_pair.db", "kernel_stats_train_pair.csv",
             "python bench.py --workload train --decoder-pair --steps 10 --warmup 3 --no-rocprof  (13 steps + 10 host-timing steps)")]
    for db, out, cmd in runs:
        if os.path.exists(j(SRC, db)):
            stats(j(SRC, db), j(OUT, "%s_%s" % (TAG, out)), cmd)
    if os.path.exists(j(SRC, "clock_e2e.db")):
        clock(j(SRC, "clock_e2e.db"), j(OUT, TAG + "_clock_pmc_e2e.txt"),
              "python bench.py --workload e2e --e2e-mode frame --no-side-stream --steps 4 --warmup 2 --no-rocprof")
    if os.path.exists(j(SRC, "fetch.db")) and os.path.exists(j(SRC, "write.db")):
        hbm(j(SRC, "fetch.db"), j(SRC, "write.db"), j(OUT, TAG + "_hbm_pmc.csv"))
    if os.path.exists(j(SRC, "mfma.db")):
        mfma(j(SRC, "mfma.db"), j(OUT, TAG + "_mfma_pmc.csv"))
    p = j(OUT, TAG + "_bench_n1.json")
    if os.path.exists(p):
        r = json.load(open(p))
        print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["traffic"])


if __name__ == "__main__":
    main()
