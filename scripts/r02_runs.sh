#!/bin/bash
# Round-2 measurement run (on the GPU box): bench records + rocprofv3 summaries -> gpurun_out/r2p
O=gpurun_out/r2p; mkdir -p $O
R=$GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/err.txt
python bench.py --workload e2e --steps 10 --warmup 3 > $O/bench_e2e.json 2>> $O/err.txt
python bench.py --workload e2e --steps 10 --warmup 3 --precision f16x3 > $O/bench_e2e_f16x3.json 2>> $O/err.txt
for p in ragged n1 scene; do python bench.py --pairs $p --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_pairs_$p.json 2>> $O/err.txt; done
python bench.py --frames 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_frames4.json 2>> $O/err.txt
python bench.py --samples 256 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n256.json 2>> $O/err.txt
python bench.py --frames 4 --samples 256 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_frames4_n256.json 2>> $O/err.txt
python bench.py --workload query+refine --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_query_refine.json 2>> $O/err.txt
python bench.py --workload query+refine --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_query_refine_f16x3.json 2>> $O/err.txt
python bench.py --precision f16x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f16x3.json 2>> $O/err.txt
for w in decoders embed train train-query; do python bench.py --workload $w --steps 10 --warmup 3 > $O/bench_$w.json 2>> $O/err.txt; done
python bench.py --workload decoders --precision f16x3 --steps 10 --warmup 3 > $O/bench_decoders_f16x3.json 2>> $O/err.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_rccl.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/p_f -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/p_w -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/p_m -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
cp /tmp/p_kt/r_results.db $O/kt.db; cp /tmp/p_f/r_results.db $O/fetch.db; cp /tmp/p_w/r_results.db $O/write.db; cp /tmp/p_m/r_results.db $O/mfma.db 2>/dev/null
tail -n 5 $O/err.txt
head -c 600 $O/bench_n1.json; echo; ls -la $O
