# round-4 evaluation-path records (bench.py --workload e2e): modes x frames per batch, frames pipelined
# over several streams; written under gpurun_out/r4e
O=gpurun_out/r4e; mkdir -p $O
for F in 1 4; do for M in stepwise frame graph; do
python bench.py --workload e2e --frames $F --e2e-mode $M --steps 40 --warmup 5 > $O/bench_e2e_${M}_f$F.json 2> $O/err.txt
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
python - <<PY
import json
r=json.load(open("$O/bench_e2e_frame_f4_streams2.json")); print("frame f4 streams=2", r["ms_per_step"], r["ms_per_frame"])
PY
