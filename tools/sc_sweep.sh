# sweep: lanes, then multi-round thresholds (R3_MIN R2_MIN)
run() { timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline > gpurun_out/sw.log 2>&1 || exit 1
python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sw.log').read().strip().split('\n')[-1])
print(sys.argv[1], round(d['ms_per_step'],2), round(d['host_enqueue_ms_per_step'],2))
" "$1"; }
for l in 0 4 8 12 16; do ZKDL_LANES=$l run "lanes=$l"; done
for cfg in "4294967295 4294967295" "4194304 2097152" "4294967295 1048576" "1048576 524288"; do
set -- $cfg
ZKDL_SC_R3_MIN=$1 ZKDL_SC_R2_MIN=$2 run "R3>=$1 R2>=$2"
done
