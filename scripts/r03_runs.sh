#!/bin/bash
# Round-3 measurement run (on the GPU box): bench records + rocprofv3 summaries -> gpurun_out/r3p
O=gpurun_out/r3p; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/err.txt
for M in stepwise frame graph; do for F in 1 4; do
python bench.py --workload e2e --e2e-mode $M --frames $F --steps 30 --warmup 5 > $O/bench_e2e_${M}_f$F.json 2>> $O/err.txt; done; done
for p in ragged n1 scene; do python bench.py --pairs $p --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_pairs_$p.json 2>> $O/err.txt; done
python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config2.json 2>> $O/err.txt
python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config3.json 2>> $O/err.txt
python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config4.json 2>> $O/err.txt
python bench.py --samples 256 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n256.json 2>> $O/err.txt
python bench.py --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f16x3.json 2>> $O/err.txt
for S in 2 3; do python bench.py --streams $S --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_streams$S.json 2>> $O/err.txt; done
for w in decoders embed train train-query; do python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2>> $O/err.txt; done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_rccl.json 2>> $O/err.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 1 --shard rays --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_rccl_rays.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_kr -o r -- python $R/bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_ke -o r -- python $R/bench.py --workload e2e --e2e-mode frame --steps 10 --warmup 3 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/p_m -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
cp /tmp/p_kt/r_results.db $O/kt.db; cp /tmp/p_kr/r_results.db $O/kt_refine.db; cp /tmp/p_ke/r_results.db $O/kt_e2e.db
cp /tmp/p_f/r_results.db $O/fetch.db; cp /tmp/p_w/r_results.db $O/write.db; cp /tmp/p_m/r_results.db $O/mfma.db 2>/dev/null
tail -n 5 $O/err.txt
head -c 400 $O/bench_n1.json; echo; ls -la $O
