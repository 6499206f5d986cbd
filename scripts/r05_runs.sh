#!/bin/bash
# (evaluation-path records: 2,000 frames after 300 — the first seconds of a process run 3-4 % slower, 300-step runs read
# 1.71-1.74 ms where 3,000-step runs read 1.67 on the same box)
# Round-5 measurement run (on the GPU box): bench records + rocprofv3 summaries -> gpurun_out/r5p
# (scripts/r05_collect.py then writes the summaries committed under profiles/r05_*)
O=gpurun_out/r5p; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/err.txt
for M in stepwise frame graph; do for F in 1 4; do
python bench.py --workload e2e --e2e-mode $M --frames $F --steps $( [ $F = 1 ] && echo 2000 || echo 500 ) --warmup $( [ $F = 1 ] && echo 300 || echo 80 ) $( [ $M = frame ] && [ $F = 1 ] && echo --pmc ) > $O/bench_e2e_${M}_f$F.json 2>> $O/err.txt; done; done
for S in 2 3 6; do python bench.py --workload e2e --e2e-mode frame --streams $S --steps 2000 --warmup 300 > $O/bench_e2e_frame_f1_streams$S.json 2>> $O/err.txt; done
python bench.py --workload e2e --e2e-mode graph --guard-every 32 --steps 2016 --warmup 320 > $O/bench_e2e_graph_f1_guard32.json 2>> $O/err.txt
python bench.py --workload e2e --e2e-mode frame --frames 4 --streams 2 --steps 500 --warmup 80 > $O/bench_e2e_frame_f4_streams2.json 2>> $O/err.txt
# larger batches per call (the round quantisation of every kernel fades: 8.53 rounds x frames)
for F in 8 16; do python bench.py --workload e2e --e2e-mode frame --frames $F --steps $((1600 / F)) --warmup $((160 / F)) --no-rocprof > $O/bench_e2e_frame_f$F.json 2>> $O/err.txt; done
python bench.py --workload e2e --e2e-mode frame --frames 16 --offsets selected --steps 100 --warmup 10 --no-rocprof > $O/bench_e2e_frame_f16_selected.json 2>> $O/err.txt
# variants: one stream inside the frame call; guard every 32nd frame; the offset decoder on the selected pairs only (opt-in)
python bench.py --workload e2e --e2e-mode frame --no-side-stream --steps 2000 --warmup 300 > $O/bench_e2e_frame_f1_onestream.json 2>> $O/err.txt
python bench.py --workload e2e --e2e-mode frame --guard-every 32 --steps 2016 --warmup 320 > $O/bench_e2e_frame_f1_guard32.json 2>> $O/err.txt
python bench.py --workload e2e --e2e-mode frame --offsets selected --steps 2000 --warmup 300 > $O/bench_e2e_frame_f1_selected.json 2>> $O/err.txt
for S in 3 6; do python bench.py --workload e2e --e2e-mode frame --offsets selected --streams $S --steps 2000 --warmup 300 > $O/bench_e2e_frame_f1_selected_streams$S.json 2>> $O/err.txt; done
python bench.py --workload e2e --e2e-mode frame --offsets selected --frames 4 --steps 500 --warmup 80 > $O/bench_e2e_frame_f4_selected.json 2>> $O/err.txt
for p in ragged n1 scene; do python bench.py --pairs $p --steps $( [ $p = ragged ] && echo 100 || echo 2000 ) --warmup $( [ $p = ragged ] && echo 20 || echo 300 ) --no-cpu-baseline $( [ $p = scene ] && echo --pmc ) > $O/bench_pairs_$p.json 2>> $O/err.txt; done
python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config2.json 2>> $O/err.txt
python bench.py --config 3 --steps 40 --warmup 10 --no-cpu-baseline --pmc > $O/bench_config3.json 2>> $O/err.txt
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config4.json 2>> $O/err.txt
python bench.py --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f16x3.json 2>> $O/err.txt
python bench.py --imnet-gf 128 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gf128.json 2>> $O/err.txt
python bench.py --imnet-gf 32 --steps 5 --warmup 2 --no-cpu-baseline --no-rocprof > $O/bench_gf32.json 2>> $O/err.txt
for w in decoders embed train train-query train-refine; do python bench.py --workload $w --steps $( [ $w = embed ] && echo 2000 || echo 40 ) --warmup $( [ $w = embed ] && echo 300 || echo 10 ) > $O/bench_$w.json 2>> $O/err.txt; done
python bench.py --offsets selected --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_selected.json 2>> $O/err.txt
python bench.py --workload train-query --offsets selected --steps 40 --warmup 10 > $O/bench_train-query_selected.json 2>> $O/err.txt
python bench.py --workload train-query --dense-offset-grad --steps 40 --warmup 10 > $O/bench_train-query_dense.json 2>> $O/err.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29654 bench.py --workload e2e --gpus 1 --steps 100 --warmup 10 --no-rocprof > $O/bench_e2e_rccl_n1.json 2>> $O/err.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_rccl.json 2>> $O/err.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 1 --shard rays --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_rccl_rays.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-rocprof > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_kr -o r -- python $R/bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline --no-rocprof > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_ke -o r -- python $R/bench.py --workload e2e --e2e-mode frame --steps 10 --warmup 3 --no-rocprof > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_tr -o r -- python $R/bench.py --workload train-refine --steps 10 --warmup 3 --no-rocprof > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_tq -o r -- python $R/bench.py --workload train-query --steps 10 --warmup 3 --no-rocprof > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_tn -o r -- python $R/bench.py --workload train --steps 10 --warmup 3 --no-rocprof > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/p_m -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-rocprof > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/p_ce -o r -- python $R/bench.py --workload e2e --e2e-mode frame --steps 4 --warmup 2 --no-rocprof > /dev/null 2>&1
cd $R
cp /tmp/p_kt/r_results.db $O/kt.db; cp /tmp/p_kr/r_results.db $O/kt_refine.db; cp /tmp/p_ke/r_results.db $O/kt_e2e.db
cp /tmp/p_tr/r_results.db $O/kt_train_refine.db; cp /tmp/p_tq/r_results.db $O/kt_train_query.db; cp /tmp/p_tn/r_results.db $O/kt_train.db
cp /tmp/p_f/r_results.db $O/fetch.db; cp /tmp/p_w/r_results.db $O/write.db; cp /tmp/p_m/r_results.db $O/mfma.db 2>/dev/null
cp /tmp/p_ce/r_results.db $O/clock_e2e.db 2>/dev/null
cp profiles/hbm_traffic.json /tmp/hbm_traffic.keep 2>/dev/null
python scripts/r05_collect.py gpurun_out/r5prof
rm -f $O/*.db
tail -n 5 $O/err.txt
head -c 300 $O/bench_n1.json; echo; ls gpurun_out/r5prof | wc -l
