"""Turn rocprofv3 (rocpd sqlite) outputs under gpurun_out/ into the small text summaries committed
under profiles/: per-kernel stats of the --kernel-trace --stats run, and per-kernel HBM byte
counters of the separate --pmc FETCH_SIZE / --pmc WRITE_SIZE runs.

usage: python scripts/prof_summary.py <tag> <kernel_trace.db> [<fetch.db> <write.db> [<mfma_pmc.db>]]
"""
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    return rows, tot


def counter(db, name):
    cur = sqlite3.connect(db).cursor()
    return list(cur.execute(
        "select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
        "where counter_name=? group by kernel_name order by sum(value) desc", (name,)))


def main():
    tag, kt = sys.argv[1], sys.argv[2]
    if len(sys.argv) == 4 and not sys.argv[3].endswith(".db"):   # <tag> <kt.db> "<command line>": stats only
        rows, tot = kernel_stats(kt)
        with open(os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag), "w") as f:
            f.write("# rocprofv3 --kernel-trace --stats -- %s\n" % sys.argv[3])
            f.write("name,calls,total_ns,avg_ns,min_ns,max_ns,pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,grid_x,wg_x\n")
            for r in rows:
                f.write('"%s",%d,%d,%.1f,%d,%d,%.3f,%s,%s,%s,%s,%s,%s,%s\n' % (
                    r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10],
                    r[11], r[12]))
        return
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    rows, tot = kernel_stats(kt)
    with open(os.path.join(out, "%s_kernel_stats.csv" % tag), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline\n")
        f.write("name,calls,total_ns,avg_ns,min_ns,max_ns,pct,vgpr,agpr,sgpr,lds_bytes,scratch_bytes,grid_x,wg_x\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.3f,%s,%s,%s,%s,%s,%s,%s\n' % (
                r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10],
                r[11], r[12]))
    print(open(os.path.join(out, "%s_kernel_stats.csv" % tag)).read())
    if len(sys.argv) >= 5:
        fetch = counter(sys.argv[3], "FETCH_SIZE")
        write = counter(sys.argv[4], "WRITE_SIZE")
        wmap = {r[0]: r for r in write}
        traffic = {}
        with open(os.path.join(out, "%s_hbm_pmc.csv" % tag), "w") as f:
            f.write("# separate passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE -- python bench.py --steps 3 --warmup 1\n")
            f.write("# values are KiB per dispatch as reported; MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE\n")
            f.write("# under-counts wide coalesced streaming reads by 2x (64 B tallied per 128 B request), so\n")
            f.write("# corrected = 2*FETCH_SIZE + WRITE_SIZE is an upper bound for mixed access widths; WRITE_SIZE uncalibrated.\n")
            f.write("kernel,dispatches,fetch_kib_avg,write_kib_avg,hbm_bytes_reported,hbm_bytes_corrected\n")
            for r in fetch:
                w = wmap.get(r[0], (r[0], 0, 0.0, 0, 0))
                rep = (r[2] + w[2]) * 1024
                cor = (2 * r[2] + w[2]) * 1024
                f.write('"%s",%d,%.1f,%.1f,%.0f,%.0f\n' % (r[0], r[1], r[2], w[2], rep, cor))
                traffic[r[0]] = {"reported": rep, "corrected": cor}
        print(open(os.path.join(out, "%s_hbm_pmc.csv" % tag)).read())
        for k, v in traffic.items():
            if "lidf_points_fused_kernel" in k:
                json.dump({"source": "profiles/%s_hbm_pmc.csv" % tag,
                           "lidf_points_kernel_bytes_per_launch": v["corrected"],
                           "reported_uncorrected": v["reported"]},
                          open(os.path.join(out, "hbm_traffic.json"), "w"))


def mfma_counters(tag, db):
    """Per-kernel averages of the MFMA counters of a separate --pmc pass."""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                            "where kernel_name like '%lidf_points%' group by kernel_name, counter_name"))
    with open(os.path.join(ROOT, "profiles", "%s_mfma_pmc.csv" % tag), "w") as f:
        f.write("# rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -- "
                "python bench.py --steps 3 --warmup 1 --no-cpu-baseline\n")
        f.write("kernel,counter,dispatches,avg_value\n")
        for r in rows:
            f.write('"%s",%s,%d,%.6g\n' % (r[0], r[1], r[2], r[3]))
    print(open(os.path.join(ROOT, "profiles", "%s_mfma_pmc.csv" % tag)).read())


if __name__ == "__main__":
    main()
    if len(sys.argv) >= 6 and sys.argv[3].endswith(".db"):
        mfma_counters(sys.argv[1], sys.argv[5])
