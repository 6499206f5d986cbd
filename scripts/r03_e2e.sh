# round-3 evaluation-path records (bench.py --workload e2e): modes x frames per batch, and frames
# pipelined over several streams; configs[3] with its stage-2 roofline
O=gpurun_out/r3e; mkdir -p $O
for F in 1 4; do for M in stepwise frame graph; do
python bench.py --workload e2e --frames $F --e2e-mode $M --steps 30 --warmup 5 > $O/bench_e2e_${M}_f$F.json 2> $O/err.txt
python - <<PY
import json; r=json.load(open("$O/bench_e2e_${M}_f$F.json")); print("$M frames=$F", r["ms_per_step"], "ms/step", r["ms_per_frame"], "ms/frame", r["stage_ms"])
PY
done; done
for S in 2 3; do python bench.py --workload e2e --e2e-mode frame --streams $S --steps 60 --warmup 6 > $O/bench_e2e_frame_f1_streams$S.json 2>> $O/err.txt
python - <<PY
import json; r=json.load(open("$O/bench_e2e_frame_f1_streams$S.json")); print("frame streams=$S", r["ms_per_step"], r["frames_per_s"])
PY
done
python bench.py --workload e2e --e2e-mode frame --frames 4 --streams 2 --steps 30 --warmup 6 > $O/bench_e2e_frame_f4_streams2.json 2>> $O/err.txt
python bench.py --config 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config3.json 2>> $O/err.txt
python - <<PY
import json; r=json.load(open("$O/bench_config3.json")); print("config3", r["ms_per_step"], r["roofline_stage2"]["pointnet"]["ms"], r["roofline_stage2"]["ief_rows"]["kernel_ms"])
r=json.load(open("$O/bench_e2e_frame_f4_streams2.json")); print("frame f4 streams=2", r["ms_per_step"], r["ms_per_frame"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_kr -o r -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_ke -o r -- python $GRAFT_REPO_ROOT/bench.py --workload e2e --e2e-mode frame --steps 10 --warmup 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; cp /tmp/p_kr/r_results.db $O/kt_refine.db; cp /tmp/p_ke/r_results.db $O/kt_e2e.db
