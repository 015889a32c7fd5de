run() { echo -n "$* $EXTRA : "; env "$@" timeout -k 10 200 python bench.py --cpu-baseline off --no-msm-probe $EXTRA 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(round(d[\"ms_per_step\"],3), {n[4:]:round(k[n]['ms_per_step'],2) for n in ('msm_accum','msm_combine','msm_bucket_reduce','msm_horner','msm_derive')})"; }
for c in 8 9 10 11 12 13; do
EXTRA="--steps 100 --warmup 5 --msm-window $c"
run A=1
done
