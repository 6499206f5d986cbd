#!/bin/bash
# Round-6 measurement run (on the GPU box): bench records + rocprofv3 summaries -> gpurun_out/r6p
# (scripts/r06_collect.py then writes the summaries committed under profiles/r06_*; scripts/verify_records.py checks
# that every fraction of a record follows from the committed CSVs alone).
# usage: r06_runs.sh [part ...]   parts: head e2e train misc prof (default: all)
O=gpurun_out/r6p; mkdir -p $O
R=$GRAFT_REPO_ROOT
PARTS="${@:-head e2e train misc prof}"
has() { [[ " $PARTS " == *" $1 "* ]]; }
B="python bench.py"
if has head; then
  $B --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/err.txt
  $B --config 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config2.json 2>> $O/err.txt
  $B --config 3 --steps 40 --warmup 10 --no-cpu-baseline --pmc > $O/bench_config3.json 2>> $O/err.txt
  $B --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config4.json 2>> $O/err.txt
  for p in ragged n1 scene; do $B --pairs $p --steps $( [ $p = ragged ] && echo 100 || echo 2000 ) --warmup $( [ $p = ragged ] && echo 20 || echo 300 ) --no-cpu-baseline $( [ $p = scene ] && echo --pmc ) > $O/bench_pairs_$p.json 2>> $O/err.txt; done
  $B --offsets selected --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_selected.json 2>> $O/err.txt
fi
if has e2e; then
  # (evaluation-path records: 2,000 frames after 300 — the first seconds of a process run 3-4 % slower)
  for M in stepwise frame graph; do for F in 1 4; do
    $B --workload e2e --e2e-mode $M --frames $F --steps $( [ $F = 1 ] && echo 2000 || echo 500 ) --warmup $( [ $F = 1 ] && echo 300 || echo 80 ) $( [ $M = frame ] && [ $F = 1 ] && echo --pmc ) > $O/bench_e2e_${M}_f$F.json 2>> $O/err.txt; done; done
  $B --workload e2e --e2e-mode frame --no-side-stream --steps 2000 --warmup 300 --pmc > $O/bench_e2e_frame_f1_onestream.json 2>> $O/err.txt
  for S in 3 6; do $B --workload e2e --e2e-mode frame --streams $S --steps 2000 --warmup 300 > $O/bench_e2e_frame_f1_streams$S.json 2>> $O/err.txt; done
  $B --workload e2e --e2e-mode frame --frames 4 --streams 2 --steps 500 --warmup 80 > $O/bench_e2e_frame_f4_streams2.json 2>> $O/err.txt
  for F in 8 16; do $B --workload e2e --e2e-mode frame --frames $F --steps $((1600 / F)) --warmup $((160 / F)) --no-rocprof > $O/bench_e2e_frame_f$F.json 2>> $O/err.txt; done
  $B --workload e2e --e2e-mode frame --offsets selected --steps 2000 --warmup 300 > $O/bench_e2e_frame_f1_selected.json 2>> $O/err.txt
  $B --workload e2e --e2e-mode frame --offsets selected --streams 6 --steps 2000 --warmup 300 > $O/bench_e2e_frame_f1_selected_streams6.json 2>> $O/err.txt
  $B --workload e2e --e2e-mode frame --frames 16 --offsets selected --steps 100 --warmup 10 --no-rocprof > $O/bench_e2e_frame_f16_selected.json 2>> $O/err.txt
  # the same stream under a 1-rank RCCL group, the SAME protocol
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29654 bench.py --workload e2e --gpus 1 --steps 2000 --warmup 300 --no-rocprof > $O/bench_e2e_rccl_n1.json 2>> $O/err.txt
fi
if has train; then
  for w in train train-query train-refine; do $B --workload $w --steps 40 --warmup 10 > $O/bench_$w.json 2>> $O/err.txt; done
  $B --workload train --decoder-pair --steps 40 --warmup 10 > $O/bench_train_pair.json 2>> $O/err.txt
  $B --workload train-query --offsets selected --steps 40 --warmup 10 > $O/bench_train-query_selected.json 2>> $O/err.txt
  $B --workload train-query --dense-offset-grad --steps 40 --warmup 10 > $O/bench_train-query_dense.json 2>> $O/err.txt
fi
if has misc; then
  $B --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f16x3.json 2>> $O/err.txt
  $B --imnet-gf 128 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gf128.json 2>> $O/err.txt
  $B --imnet-gf 32 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gf32.json 2>> $O/err.txt
  LIDF_CHAIN16=0 $B --imnet-gf 128 --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof > $O/bench_gf128_layers.json 2>> $O/err.txt
  LIDF_CHAIN16=0 $B --imnet-gf 32 --steps 5 --warmup 2 --no-cpu-baseline --no-rocprof > $O/bench_gf32_layers.json 2>> $O/err.txt
  for w in decoders embed; do $B --workload $w --steps $( [ $w = embed ] && echo 2000 || echo 40 ) --warmup $( [ $w = embed ] && echo 300 || echo 10 ) > $O/bench_$w.json 2>> $O/err.txt; done
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_rccl.json 2>> $O/err.txt
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 1 --shard rays --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_rccl_rays.json 2>> $O/err.txt
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  H="--steps 10 --warmup 2 --no-cpu-baseline --no-rocprof --no-split-f16"
  P="--steps 3 --warmup 1 --no-cpu-baseline --no-rocprof --no-split-f16"
  kt() { rm -rf /tmp/p_$1; rocprofv3 --kernel-trace --stats -d /tmp/p_$1 -o r -- python $R/bench.py "${@:2}" > /dev/null 2>&1; cp /tmp/p_$1/r_results.db $R/$O/$1.db 2>/dev/null || cp $(find /tmp/p_$1 -name '*_results.db' | head -1) $R/$O/$1.db; }
  kt kt $H
  kt kt_refine --config 3 $H
  kt kt_e2e --workload e2e --e2e-mode frame --steps 10 --warmup 3 --no-rocprof
  kt kt_e2e_onestream --workload e2e --e2e-mode frame --no-side-stream --steps 10 --warmup 3 --no-rocprof
  kt kt_train_refine --workload train-refine --steps 10 --warmup 3 --no-rocprof
  kt kt_train_query --workload train-query --steps 10 --warmup 3 --no-rocprof
  kt kt_train --workload train --steps 10 --warmup 3 --no-rocprof
  kt kt_train_pair --workload train --decoder-pair --steps 10 --warmup 3 --no-rocprof
  pmc() { rm -rf /tmp/p_$1; rocprofv3 "${@:3}" -d /tmp/p_$1 -o r -- python $R/bench.py $2 > /dev/null 2>&1; cp $(find /tmp/p_$1 -name '*_results.db' | head -1) $R/$O/$1.db; }
  pmc fetch "$P" --pmc FETCH_SIZE
  pmc write "$P" --pmc WRITE_SIZE
  pmc mfma "$P" --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
  pmc clock_e2e "--workload e2e --e2e-mode frame --no-side-stream --steps 4 --warmup 2 --no-rocprof" --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
  cd $R
  python scripts/r06_collect.py gpurun_out/r6prof
  python scripts/verify_records.py r06 gpurun_out/r6prof > gpurun_out/r6prof/r06_verify.txt 2>&1; tail -4 gpurun_out/r6prof/r06_verify.txt
  rm -f $O/*.db
fi
tail -n 5 $O/err.txt
head -c 300 $O/bench_n1.json; echo; ls gpurun_out/r6prof 2>/dev/null | wc -l
