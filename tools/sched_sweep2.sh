run() { echo -n "$* $EXTRA : "; env "$@" timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"ms_per_step\"],3))"; }
EXTRA="--steps 100 --warmup 5"
run ZKDL_HP_GROUP=1
run ZKDL_HP_GROUP=4
run ZKDL_HP_GROUP=0
run ZKDL_HP_GROUP=3
run ZKDL_HP_GROUP=4 ZKDL_EXPERIMENT_SKIP_G1=1
run ZKDL_HP_GROUP=1 ZKDL_EXPERIMENT_SKIP_G1=1
