#!/bin/bash
# Secondary bench records of round 2 (run on the GPU box; outputs under gpurun_out/r2b)
O=gpurun_out/r2b; mkdir -p $O
python bench.py --workload e2e --steps 10 --warmup 3 > $O/bench_e2e.json 2> $O/e2e.err
python bench.py --workload e2e --steps 10 --warmup 3 --precision f16x3 > $O/bench_e2e_f16x3.json 2>> $O/e2e.err
for p in ragged n1 scene; do python bench.py --pairs $p --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_pairs_$p.json 2>> $O/pairs.err; done
python bench.py --frames 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_frames4.json 2> $O/f4.err
python bench.py --samples 256 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n256.json 2> $O/n256.err
python bench.py --workload query+refine --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_query_refine.json 2> $O/qr.err
tail -n 3 $O/*.err
cat $O/*.json
