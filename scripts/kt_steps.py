"""Per-step view of a rocprofv3 kernel trace (rocpd sqlite): launches and busy time per step, and the kernels of
a step largest first. usage: python scripts/kt_steps.py <results.db> <steps incl. warm-up> [top]"""
import sqlite3
import sys

db, nstep = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
cur = sqlite3.connect(db).cursor()
per = {}
for name, dur in cur.execute("select name, duration from kernels"):
    k = name.split("(")[0].replace("void ", "").strip()
    per.setdefault(k, []).append(dur)
tot_l = sum(len(v) for v in per.values())
tot_t = sum(sum(v) for v in per.values())
print("launches/step %.1f   busy ms/step %.3f   (%d steps)" % (tot_l / nstep, tot_t / nstep / 1e6, nstep))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print("%7.1f us/step  %5.1f x %8.1f us  %s" % (sum(v) / nstep / 1e3, len(v) / nstep, sum(v) / len(v) / 1e3, k[:100]))
