import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/proof", round(d["ms_per_step"], 3), "agg", d["aggregate_fr"][:14])
for n, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_per_step"])[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print("  %-24s %.3f ms  conc %.3f" % (n, v["ms_per_step"], v.get("concurrent_ms_per_step") or 0))
