run() { echo -n "$* $EXTRA : "; env "$@" timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"ms_per_step\"],3), d['kernels']['sumcheck_hp_round'])"; }
EXTRA="--steps 100 --warmup 5"
run ZKDL_NO_HP0=1
run A=1
run ZKDL_NO_HP0=1
run A=1
