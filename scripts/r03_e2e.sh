# round-3 evaluation-path records (bench.py --workload e2e) in the three modes, 1 and 4 frames
O=gpurun_out/r3e; mkdir -p $O
for F in 1 4; do for M in stepwise frame graph; do
python bench.py --workload e2e --frames $F --e2e-mode $M --steps 30 --warmup 5 > $O/e2e_${M}_f$F.json 2> $O/e2e_${M}_f$F.err
python - <<PY
import json; r=json.load(open("$O/e2e_${M}_f$F.json")); print("$M frames=$F", r["ms_per_step"], "ms/step", r["ms_per_frame"], "ms/frame", r["stage_ms"])
PY
done; done
