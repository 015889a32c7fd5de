run() { echo -n "$* $EXTRA : "; env "$@" timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d[\"ms_per_step\"],3))"; }
zkdl_amd/lib/gpu_touch; echo touch rc=$?
EXTRA="--steps 20 --warmup 5"
run A=after_touch
run A=second
