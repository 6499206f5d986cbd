"""Round-5 profile summaries: gpurun_out/r5p (scripts/r05_runs.sh) -> profiles/r05_*: the bench records as
they were printed, per-kernel statistics of the kernel-trace runs (headline, configs[3], evaluation path,
stage-2 training step), the HBM and matrix-pipe counters of the headline, clock / pipe occupancy of the
evaluation path's kernels."""
import json
import os
import shutil
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r5p")
# run on the GPU box (the rocprofv3 databases are too large to travel back): summaries land in
# gpurun_out/r5prof, which is then copied into profiles/
OUT = os.path.join(ROOT, sys.argv[1]) if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
os.makedirs(OUT, exist_ok=True)


def stats(db, out, cmd):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % cmd)
        f.write("name,calls,total_ns,avg_ns,min_ns,max_ns,pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,grid_x,wg_x\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.3f,%s,%s,%s,%s,%s,%s,%s\n' % (
                r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))


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


def main():
    for f in sorted(os.listdir(SRC)):
        if f.startswith("bench_") and f.endswith(".json") and os.path.getsize(os.path.join(SRC, f)) > 0:
            shutil.copy(os.path.join(SRC, f), os.path.join(OUT, "r05_" + f))
    j = os.path.join
    stats(j(SRC, "kt.db"), j(OUT, "r05_kernel_stats.csv"), "python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rocprof")
    stats(j(SRC, "kt_refine.db"), j(OUT, "r05_kernel_stats_refine.csv"),
          "python bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-rocprof")
    stats(j(SRC, "kt_e2e.db"), j(OUT, "r05_kernel_stats_e2e.csv"),
          "python bench.py --workload e2e --e2e-mode frame --steps 10 --warmup 3 --no-rocprof")
    stats(j(SRC, "kt_train_refine.db"), j(OUT, "r05_kernel_stats_train_refine.csv"),
          "python bench.py --workload train-refine --steps 10 --warmup 3 --no-rocprof  (13 steps)")
    if os.path.exists(j(SRC, "kt_train_query.db")):
        stats(j(SRC, "kt_train_query.db"), j(OUT, "r05_kernel_stats_train_query.csv"),
              "python bench.py --workload train-query --steps 10 --warmup 3 --no-rocprof  (13 steps)")
    if os.path.exists(j(SRC, "kt_train.db")):
        stats(j(SRC, "kt_train.db"), j(OUT, "r05_kernel_stats_train.csv"),
              "python bench.py --workload train --steps 10 --warmup 3 --no-rocprof  (13 steps)")
    if os.path.exists(j(SRC, "clock_e2e.db")):
        clock(j(SRC, "clock_e2e.db"), j(OUT, "r05_clock_pmc_e2e.txt"),
              "python bench.py --workload e2e --e2e-mode frame --steps 4 --warmup 2 --no-rocprof")
    # headline counters through the round-2 summariser (HBM: corrected 2 x FETCH + WRITE; matrix-pipe counters)
    subprocess.run([sys.executable, j(ROOT, "scripts", "prof_summary.py"), "r05tmp", j(SRC, "kt.db"), j(SRC, "fetch.db"),
                    j(SRC, "write.db"), j(SRC, "mfma.db")], check=True, stdout=subprocess.DEVNULL)
    P = j(ROOT, "profiles")
    os.replace(j(P, "r05tmp_hbm_pmc.csv"), j(OUT, "r05_hbm_pmc.csv"))
    os.replace(j(P, "r05tmp_mfma_pmc.csv"), j(OUT, "r05_mfma_pmc.csv"))
    os.remove(j(P, "r05tmp_kernel_stats.csv"))
    r = json.load(open(j(OUT, "r05_bench_n1.json")))
    print("headline", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["traffic"])


if __name__ == "__main__":
    main()
