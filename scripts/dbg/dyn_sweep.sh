for cfg in "0 0" "4 1" "4 2" "2 1" "8 1" "8 2" "1 1"; do set -- $cfg
 LIDF_DYN_MIN=$1 LIDF_DYN_CHUNK=$2 python bench.py --pairs scene --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('scene min=$1 chunk=$2', r['roofline']['kernel_ms'], r['roofline']['frac'], r['ms_per_step'])"
done
for cfg in "0 0" "8 1" "4 1"; do set -- $cfg
 LIDF_DYN_MIN=$1 LIDF_DYN_CHUNK=$2 python bench.py --pairs ragged --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('ragged min=$1 chunk=$2', r['roofline']['kernel_ms'], r['roofline']['frac'], r['ms_per_step'])"
 LIDF_DYN_MIN=$1 LIDF_DYN_CHUNK=$2 python bench.py --pairs n1 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('n1 min=$1 chunk=$2', r['roofline']['kernel_ms'], r['roofline']['frac'], r['ms_per_step'])"
done
