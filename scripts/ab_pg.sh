#!/bin/bash
# Why is a frame of the evaluation stream slower under a process group (1.97 ms per frame under a
# 1-rank nccl group against 1.66 without one)? One gpurun session, the SAME protocol for every leg (2,000 frames after
# 300; the r05 record ran 100 after 10), each leg twice, alternating:
#   plain          python bench.py --workload e2e                         (no process group)
#   nccl           torchrun, 1 rank, backend nccl (= RCCL)                (what the driver's N = 1 launch is)
#   nccl-onestream the same with --no-side-stream                         (is it the side stream's queue mapping?)
#   gloo           torchrun, 1 rank, backend gloo (LIDF_TEST_SHARE_GPU)   (is it RCCL, or any process group?)
#   nccl-omp       nccl with OMP_NUM_THREADS unset by hand (torchrun exports 1)
#   nccl-hwq8      nccl with GPU_MAX_HW_QUEUES=8
#   plain-omp1     no process group, OMP_NUM_THREADS=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_pg; mkdir -p $O
N=${1:-2000}; W=${2:-300}
run_plain() { python $R/bench.py --workload e2e --e2e-mode frame --steps $N --warmup $W --no-rocprof "$@"; }
run_tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) $R/bench.py --workload e2e --gpus 1 --steps $N --warmup $W --no-rocprof "$@"; }
for rep in 1 2; do
  run_plain > $O/plain_$rep.json 2>/dev/null
  run_tr > $O/nccl_$rep.json 2>/dev/null
  run_tr --no-side-stream > $O/nccl-onestream_$rep.json 2>/dev/null
  run_plain --no-side-stream > $O/plain-onestream_$rep.json 2>/dev/null
  LIDF_TEST_SHARE_GPU=1 run_tr > $O/gloo_$rep.json 2>/dev/null
  OMP_NUM_THREADS=16 run_tr > $O/nccl-omp16_$rep.json 2>/dev/null
  GPU_MAX_HW_QUEUES=8 run_tr > $O/nccl-hwq8_$rep.json 2>/dev/null
  OMP_NUM_THREADS=1 run_plain > $O/plain-omp1_$rep.json 2>/dev/null
done
python - <<'PY'
import glob, json, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "ab_pg")
print("%-22s %10s %10s %10s %10s" % ("leg", "ms/frame", "frame", "metrics", "all_gather"))
for f in sorted(glob.glob(O + "/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        s = r.get("stage_ms", {})
        print("%-22s %10.4f %10.4f %10.4f %10.4f" % (os.path.basename(f)[:-5], r["ms_per_frame"], s.get("frame", 0), s.get("metrics", 0), s.get("all_gather", 0)))
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
