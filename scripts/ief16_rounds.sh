# Duration of lidf_ief16_kernel<2> over 1..4 frames per call (4.69 / 9.375 / 14.06 / 18.75 sixteen-ray sub-tiles per SIMD):
# what the partial last round of a SIMD's two wavefronts costs (one wavefront alone on the odd sub-tile)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in 1 2 3 4; do
  rm -rf /tmp/p_$f; rocprofv3 --kernel-trace --stats -d /tmp/p_$f -o r -- python $R/bench.py --workload query+refine --frames $f --samples 4 --steps 12 --warmup 3 --no-rocprof --no-cpu-baseline --no-split-f16 > /tmp/o_$f.log 2>&1 || tail -3 /tmp/o_$f.log
  python - <<PY
import sqlite3
cur=sqlite3.connect('/tmp/p_$f/r_results.db').cursor()
print("frames $f", " | ".join("%s x%d avg %.1f min %.1f us"%(r[0][:40], r[1], r[2]/1e3, r[3]/1e3) for r in cur.execute("select name,count(*),avg(duration),min(duration) from kernels where name like '%ief16%' or name like '%pointnet_chain%' group by name")))
PY
done
