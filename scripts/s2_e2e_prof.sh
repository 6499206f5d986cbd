O=gpurun_out/s2; mkdir -p $O; R=$GRAFT_REPO_ROOT
python bench.py --workload e2e --steps 20 --warmup 3 > $O/e2e_1.json 2> $O/err.txt
python bench.py --workload e2e --frames 4 --steps 10 --warmup 3 > $O/e2e_4.json 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_e -o r -- python $R/bench.py --workload e2e --steps 20 --warmup 3 > /dev/null 2>&1
cd $R; cp /tmp/p_e/r_results.db $O/e2e_kt.db
cat $O/e2e_1.json $O/e2e_4.json | cut -c1-1500
