#!/usr/bin/env python3
"""BASELINE configs[4]: the 2^24-point MSM stress with the Pippenger window-width sweep, on one GPU, through the C ABI.
4096 generators x 4096 rows = 2^24 (point, scalar) pairs, the shape of one width-4096 layer's commitment:
  * zk_msm_batch with FULL-WIDTH (uniform 255-bit) scalars, and
  * zk_commit with quantised-weight scalars (|w| < 2^13: what Commitment::commit sees, commitment.cu:29-41),
HIP-event timed, window width c = 6 .. 13 (0 = the library's choice).  Prints a table; --json writes it as JSON."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_free_gen import generator_jac  # noqa: E402
from zkdl_amd.capi import MSM_PIPPENGER, Zk  # noqa: E402


def main():
    zk = Zk(0)
    zk.set_msm_mode(MSM_PIPPENGER)
    zk.seed(11)
    n = int(sys.argv[sys.argv.index("--gens") + 1]) if "--gens" in sys.argv else 4096          # generators per MSM; rows = 2^24 / n
    q = (1 << 24) // n
    bits = int(sys.argv[sys.argv.index("--bits") + 1]) if "--bits" in sys.argv else 0             # 0: uniform below r; else uniform below 2^bits (an empty top window)
    gens = zk.gens(zk.g1_mul(zk.g1_fill(n, generator_jac()), zk.fr_random(n)))
    if bits:
        import numpy as np
        raw = np.random.default_rng(5).integers(0, 1 << 32, size=(n * q, 8), dtype=np.uint64).astype(np.uint32)
        for limb in range(8):
            keep = min(32, max(0, bits - 32 * limb))
            raw[:, limb] &= np.uint32((1 << keep) - 1 if keep < 32 else 0xffffffff)
        full = zk.upload(raw, 8)
    else:
        full = zk.fr_random(n * q)
    small = zk.fr_mont(zk.fr_random_int(n * q, 13))
    rows = []
    ref_full = ref_small = None
    widths = [int(x) for x in sys.argv[sys.argv.index("--windows") + 1].split(",")] if "--windows" in sys.argv else [0] + list(range(6, 14))
    for c in widths:          # zk_set_msm_window admits 2 .. 13 (the r03 sweep listed 14-16: those were 13, clamped)
        zk.set_msm_window(c)
        rec = {"window": c}
        for name, fn in (("fullwidth_msm_batch", lambda: zk.msm_batch(gens, full)), ("quantised_commit", lambda: zk.commit(gens, small))):
            out = fn()                      # builds this width's window table, warms the pools
            zk.sync()
            zk.timer_start()
            out = fn()
            ms = zk.timer_stop_ms()
            rec[name + "_ms"] = ms
            rec[name + "_mops"] = n * q / ms / 1e3
            if "--phases" in sys.argv and name.startswith("full"):   # the same call once more under the per-launch event profile (serialises nothing: one stream)
                zk.prof_enable(True)
                zk.prof_reset()
                out = fn()
                zk.sync()
                rec["phases_ms"] = {k: round(v["ms"], 3) for k, v in zk.prof_read().items() if v["ms"] > 0.05}
                zk.prof_enable(False)
                print("      phases:", "  ".join("%s=%.2f" % kv for kv in sorted(rec["phases_ms"].items(), key=lambda kv: -kv[1])), flush=True)
            aff, inf = zk.g1_to_affine(out.view(0, 64))          # every width must give the same group elements
            key = (aff.tobytes(), inf.tobytes())
            if name.startswith("full"):
                ref_full = ref_full or key
                assert key == ref_full, "window %d changes the MSM result" % c
            else:
                ref_small = ref_small or key
                assert key == ref_small, "window %d changes the commitment" % c
            del out
        rows.append(rec)
        print("c=%2d   full-width 2^24 pairs: %8.2f ms  %7.1f Mpairs/s     quantised commit: %7.2f ms  %7.1f Mpairs/s" %
              (c, rec["fullwidth_msm_batch_ms"], rec["fullwidth_msm_batch_mops"], rec["quantised_commit_ms"], rec["quantised_commit_mops"]), flush=True)
    # the bucket-free path: every signed digit of every window precomputed (zk_gens_prepare_full); same group elements
    zk.set_msm_window(0)
    table_rows = []
    for c in (() if "--no-tables" in sys.argv else (8, 10, 11, 12, 13)):
        g2 = zk.gens(gens.points())
        g2.prepare_full(c)
        zk.sync()
        out = zk.msm_batch(g2, full)
        zk.sync()
        zk.timer_start()
        out = zk.msm_batch(g2, full)
        ms = zk.timer_stop_ms()
        aff, inf = zk.g1_to_affine(out.view(0, 64))
        assert (aff.tobytes(), inf.tobytes()) == ref_full, "digit tables of width %d change the MSM result" % c
        table_gb = 128.0 * n * ((256 + c - 1) // c) * (1 << (c - 1)) / 1e9
        table_rows.append({"window": c, "fullwidth_msm_batch_ms": ms, "fullwidth_msm_batch_mops": n * q / ms / 1e3, "table_gb": table_gb})
        print("digit tables c=%2d (%.1f GB)   full-width 2^24 pairs: %8.2f ms  %7.1f Mpairs/s" % (c, table_gb, ms, n * q / ms / 1e3), flush=True)
        del out
        g2.close()
    if "--json" in sys.argv:
        json.dump({"pairs": n * q, "n_gens": n, "rows": q, "sweep": rows, "digit_tables": table_rows}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    zk.close()


if __name__ == "__main__":
    main()
