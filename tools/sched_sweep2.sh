run() { echo -n "$* $EXTRA : "; env "$@" timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe --steps 150 --warmup 12 $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"ms_per_step\"],3), 'host', round(d['host_enqueue_ms_per_step'],2))"; }
EXTRA="--inflight 8"
run ZKDL_EXPERIMENT_SKIP_G1=1 ZKDL_G1_STREAMS=2
EXTRA="--inflight 3"
run ZKDL_EXPERIMENT_SKIP_G1=1 ZKDL_G1_STREAMS=6
EXTRA="--inflight 2"
run ZKDL_EXPERIMENT_SKIP_G1=1 ZKDL_G1_STREAMS=2
EXTRA="--inflight 4"
run ZKDL_EXPERIMENT_SKIP_G1=1 ZKDL_G1_STREAMS=2
EXTRA="--inflight 3"
run ZKDL_EXPERIMENT_SKIP_G1=1 ZKDL_G1_STREAMS=1
