#!/bin/bash
# A/B of the per-point kernel's net-by-net last round (LIDF_TAIL_SPLIT=0 keeps the tiles of the partial round
# whole) and of the library before the change (liblidf_old.so at the repo root), in ONE gpurun call:
# box-to-box variance is 2-3 %. Alternating runs; the JSON records land in gpurun_out/ab_tail.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_tail; mkdir -p $O
cd $R
for rep in 1 2; do
for v in split whole old; do
  unset LIDF_HIP_LIB LIDF_TAIL_SPLIT
  [ $v = whole ] && export LIDF_TAIL_SPLIT=0
  [ $v = old ] && [ -f $R/liblidf_old.so ] && export LIDF_HIP_LIB=$R/liblidf_old.so
  python bench.py --pairs scene --steps 200 --warmup 20 --no-cpu-baseline --no-rocprof > $O/scene_${v}_$rep.json 2>/dev/null
  python bench.py --workload e2e --e2e-mode frame --steps 300 --warmup 30 --no-rocprof > $O/e2e_f1_${v}_$rep.json 2>/dev/null
  python bench.py --workload e2e --e2e-mode frame --frames 4 --steps 80 --warmup 8 --no-rocprof > $O/e2e_f4_${v}_$rep.json 2>/dev/null
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-rocprof --no-split-f16 > $O/headline_${v}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "ab_tail")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "ERR", e); continue
    print("%-28s ms/step %.4f  kernel_ms %s  frac %s" % (os.path.basename(f), r["ms_per_step"],
          (r.get("roofline") or {}).get("kernel_ms"), (r.get("roofline") or {}).get("frac")))
PY
