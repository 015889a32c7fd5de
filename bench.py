#!/usr/bin/env python3
"""bench.py -- proving ms/example at batch 256 on the 18M-parameter ModulusLab MLP (BASELINE.json
metric), one process per GPU, independent proof batches per rank, RCCL all-reduce of the per-proof
aggregate over xGMI.

A "step" is one pass of the reference's timed region (demo.cu:124-138): fcs[L-1].prove, then for
i = L-2..0 relus[i].prove + fcs[i].prove, over one synthetic batch that is already resident in HBM
(weights quantised + committed, inference done: all of that is setup in the reference too), plus the
transcript digest and, for N > 1, its all-reduce.  value = step time / (256 * N) in ms, max over
ranks, exactly K steps between barriers.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mlp18m_b256|mlp18m_b1|tiny]
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

MLP18M = [784, 1000, 1773, 1773, 1773, 1773, 1773, 1124, 1000]      # model.py:14-30
WORKLOADS = {"mlp18m_b256": (MLP18M, 256), "mlp18m_b1": (MLP18M, 1), "tiny": ([64, 96, 48, 32], 16),
             "deep4096x64_b256": ([4096] * 65, 256), "deep4096x8_b256": ([4096] * 9, 256)}   # BASELINE configs[4] (a parity/stress case, not the bench line)
R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def ceil_log2(x):
    return 0 if x <= 1 else (x - 1).bit_length()


def layer_shapes(dims):
    out = []
    for n_in, n_out in zip(dims[:-1], dims[1:]):
        in_p, out_p = 1 << ceil_log2(n_in), 1 << ceil_log2(n_out)
        n_g = 1 << ((ceil_log2(n_in * n_out) + 1) // 2)             # demo.cu:81 (unpadded in*out)
        out.append(dict(n_in=n_in, n_out=n_out, in_p=in_p, out_p=out_p, n_gens=n_g, m=in_p * out_p // n_g))
    return out


class Workload:
    """Everything the reference builds before timer.start(): generators, quantised + committed
    layers, quantised input, inference (Z_i, A_i, aux bits).  Synthetic weights in model.py's init
    distribution, synthetic N(0,1) inputs; seeded per rank so ranks prove independent batches."""

    def __init__(self, zk, dims, batch, seed, inflight=2):
        from oracle_free_gen import generator_jac
        from zkdl_amd.capi import FC, ReLU, Transcript
        self.zk, self.dims, self.batch = zk, dims, batch
        self.shapes = layer_shapes(dims)
        self.B_p = 1 << ceil_log2(batch)
        rng = np.random.default_rng(1000 + seed)
        zk.seed(0xC0FFEE + seed)
        L = len(self.shapes)
        self.gens, self.fcs, self.info = [], [], {}
        t_gen = t_commit = 0.0
        pairs = 0
        for s in self.shapes:
            # generators = G * random scalars (demo.cu:81-82)
            t0 = time.perf_counter()
            base = zk.g1_fill(s["n_gens"], generator_jac())
            g = zk.gens(zk.g1_mul(base, zk.fr_random(s["n_gens"])))
            zk.sync()
            t_gen += time.perf_counter() - t0
            self.gens.append(g)
            w = rng.uniform(-1.0 / np.sqrt(s["n_in"]), 1.0 / np.sqrt(s["n_in"]), size=(s["n_in"], s["n_out"])).astype(np.float32)
            wd = zk.upload_f32(w)
            zk.sync()
            t0 = time.perf_counter()
            self.fcs.append(FC.from_float(zk, s["n_in"], s["n_out"], wd, g))     # quantise + mont + commit
            zk.sync()
            t_commit += time.perf_counter() - t0
            pairs += s["in_p"] * s["out_p"]
        x = rng.standard_normal((batch, dims[0])).astype(np.float32)
        self.X = zk.fr_mont(zk.fc_load_input(batch, dims[0], zk.upload_f32(x)))
        self.relus = [ReLU(zk) for _ in range(L - 1)]
        self.Z, self.A = [], []
        cur = self.X
        t0 = time.perf_counter()
        for i in range(L - 1):
            z = self.fcs[i].forward(cur)
            a = self.relus[i].forward(z)
            self.Z.append(z)
            self.A.append(a)
            cur = a
        self.Yhat = zk.fr_unmont(self.fcs[L - 1].forward(cur))                   # demo.cu:119
        zk.sync()
        t_fwd = time.perf_counter() - t0
        # rotated: proof i's G1 phase runs under the Fr phases of the following proofs; a transcript is reused
        # only after its own G1 phase has finished, so with T transcripts the period is >= (Fr + G1 chain) / T
        self.trs = [Transcript(zk) for _ in range(max(2, inflight))]
        self.tr = self.trs[0]
        self.info = dict(setup_generators_s=t_gen, setup_commit_s=t_commit, commit_pairs=pairs,
                         commit_mops=pairs / t_commit / 1e6, inference_s=t_fwd)
        # algorithmic Fr bytes of one proof: sumcheck 96*S (binary) / 192*S (ip, hadamard),
        # me 32*n, partial_me 32*n*(1+2^-k)
        fr_bytes = 0
        for i, s in enumerate(self.shapes):
            xin = self.B_p * s["in_p"]
            wn = s["in_p"] * s["out_p"]
            zn = self.B_p * s["out_p"]
            fr_bytes += 32 * (xin + s["in_p"]) + 32 * (wn + s["in_p"]) + 192 * s["in_p"] + 32 * zn + 32 * (wn + s["n_gens"])
            if i < L - 1:
                fr_bytes += 96 * 32 * zn + 32 * (32 * zn + 32) + 96 * 16 * zn + 32 * (16 * zn + 16) + 192 * zn
        self.info["algorithmic_fr_bytes_per_proof"] = fr_bytes

    def challenges(self, step):
        """Fresh challenge vectors in the reference's draw order (zkfc.cu:135-137, zkrelu.cu:85-98)."""
        zk, sh, L = self.zk, self.shapes, len(self.shapes)
        kb = ceil_log2(self.B_p)

        def fc(s):
            return [zk.random_vec(kb), zk.random_vec(ceil_log2(s["in_p"])), zk.random_vec(ceil_log2(s["out_p"]))]

        def relu(s):
            k = ceil_log2(self.B_p * s["out_p"])
            return [zk.random_vec(n) for n in (k + 5, k + 5, k + 4, k + 4, k, k, k)]

        out = [("fc", L - 1, fc(sh[L - 1]))]
        for i in range(L - 2, -1, -1):
            out.append(("relu", i, relu(sh[i])))
            out.append(("fc", i, fc(sh[i])))
        return out

    def prove(self, chal, tr=None):
        """The reference's timed region, enqueued on the library's streams (no host sync inside)."""
        L = len(self.shapes)
        tr = tr or self.tr
        tr.reset()
        for kind, i, ch in chal:
            if kind == "fc":
                x_in = self.A[i - 1] if i > 0 else self.X
                z_out = self.Yhat if i == L - 1 else self.Z[i]
                self.fcs[i].prove(x_in, z_out, self.gens[i], ch[0], ch[1], ch[2], tr)
            else:
                self.relus[i].prove(self.Z[i], self.A[i], ch, tr)


from zkdl_amd.dist import lanes_to_fr  # noqa: E402


def cpu_baseline(zk, wl, max_seconds=60.0):
    """Time the oracle (CPU restatement of the reference algorithm) on a bounded sample of the same
    workload: the proving loop of the LAST TWO FC layers + the ReLU between them of the 18M MLP at
    batch 1 (weight-side work is batch-independent), OpenMP over tensor elements."""
    from oracle.oracle_py import Oracle
    orc = Oracle()
    sh = wl.shapes
    L = len(sh)
    sample_layers = [L - 1, L - 2]
    rng = np.random.default_rng(5)
    t_start = time.perf_counter()
    Wq, gens, coms = {}, {}, {}
    for i in sample_layers:
        s = sh[i]
        w = rng.uniform(-1.0 / np.sqrt(s["n_in"]), 1.0 / np.sqrt(s["n_in"]), size=(s["n_in"], s["n_out"])).astype(np.float32)
        Wq[i] = orc.fr_mont(orc.float_to_fr(w, s["in_p"], s["out_p"]))
        gens[i] = wl.gens[i].points().download()               # inputs only; any points time the same
        coms[i] = wl.fcs[i].com().download()
    x = rng.standard_normal((1, sh[L - 2]["n_in"])).astype(np.float32)
    X = orc.fr_mont(orc.float_to_fr(np.abs(x), 1, sh[L - 2]["in_p"]))
    Z = orc.fc_forward(X, Wq[L - 2], 1, sh[L - 2]["in_p"], sh[L - 2]["out_p"])
    A, sign, mag, rem = orc.relu_forward(Z)
    Y = orc.fr_unmont(orc.fc_forward(A, Wq[L - 1], 1, sh[L - 1]["in_p"], sh[L - 1]["out_p"]))
    prep = time.perf_counter() - t_start

    def rv(n):
        a = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
        a[:, 7] = (a[:, 7].astype(np.uint64) % 1944954707).astype(np.uint32)
        return a

    t0 = time.perf_counter()
    s = sh[L - 1]
    orc.fc_prove(A, Y, Wq[L - 1], s["in_p"], s["out_p"], gens[L - 1], coms[L - 1], rv(0), rv(ceil_log2(s["in_p"])), rv(ceil_log2(s["out_p"])))
    k = ceil_log2(sh[L - 2]["out_p"])
    orc.relu_prove(Z, A, sign, mag, rem, rv(k + 5), rv(k + 5), rv(k + 4), rv(k + 4), rv(k), rv(k), rv(k))
    s = sh[L - 2]
    orc.fc_prove(X, Z, Wq[L - 2], s["in_p"], s["out_p"], gens[L - 2], coms[L - 2], rv(0), rv(ceil_log2(s["in_p"])), rv(ceil_log2(s["out_p"])))
    dt = time.perf_counter() - t0
    sample_w = sum(sh[i]["in_p"] * sh[i]["out_p"] for i in sample_layers)
    total_w = sum(s_["in_p"] * s_["out_p"] for s_ in sh)
    return {"value": dt * 1e3 * total_w / sample_w, "unit": "ms/example (batch 1, extrapolated to 8 layers by padded weight count)",
            "cores": orc.num_threads(), "kind": "port",
            "sample": "oracle prove loop of FC[7] + ReLU[6] + FC[6] of the 18M MLP at batch 1: %.1f s measured (prep %.1f s untimed); "
                      "the reference's CUDA binary cannot run here (no NVIDIA device)" % (dt, prep),
            "measured_sample_ms": dt * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="mlp18m_b256", choices=sorted(WORKLOADS))
    ap.add_argument("--msm", default="pippenger", choices=["pippenger", "literal"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=3, help="transcripts rotated by the pipelined loop (proofs whose G1 phase may still be running)")
    ap.add_argument("--serial", action="store_true", help="no worker streams at all: every launch runs alone (the schedule the roofline durations are measured under)")
    ap.add_argument("--msm-window", type=int, default=0, help="force the Pippenger window width (0 = library default)")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every proof on the main stream (no overlap of proof i's G1 phase with proof i+1's Fr phase)")
    ap.add_argument("--backend", default=os.environ.get("BENCH_BACKEND", "nccl"), help="nccl (= RCCL, default) | gloo (rehearsal on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses device 0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from zkdl_amd.capi import MSM_LITERAL, MSM_PIPPENGER, Zk
    zk = Zk(local_rank)
    zk.set_msm_mode(MSM_PIPPENGER if args.msm == "pippenger" else MSM_LITERAL)
    if args.msm_window:
        zk.set_msm_window(args.msm_window)
    dims, batch = WORKLOADS[args.workload]
    wl = Workload(zk, dims, batch, seed=rank, inflight=args.inflight)

    # The G1 phase of every proof (MSM batch, canonicalisation, digest) and the all-reduce of its digest are
    # ordered on the library's G1 stream; the Fr phase of the next proof runs beside them on the worker lanes.
    pipelined = not args.no_pipeline and not args.serial
    if args.serial:
        zk.set_flag("lanes", 0)
    # one digest buffer per (transcript, G1 stream) combination: a buffer is then only ever reused by a proof that runs on
    # the same G1 stream, i.e. strictly behind the all-reduce that last read it
    aggs = [torch.zeros(8, dtype=torch.int64, device="cuda") for _ in range(4 * max(2, args.inflight))]
    ext_streams = {}
    torch.cuda.synchronize()

    def ext_stream(ptr):
        if ptr not in ext_streams:
            ext_streams[ptr] = torch.cuda.ExternalStream(ptr, device=torch.device("cuda", local_rank))
        return ext_streams[ptr]

    def step(i):
        chal = wl.challenges(i)
        tr = wl.trs[i % len(wl.trs)] if pipelined else wl.tr
        agg = aggs[i % len(aggs)]
        wl.prove(chal, tr)
        if pipelined:
            tr.seal(agg.data_ptr())
        else:
            tr.digest_dev(agg.data_ptr())
        if world > 1:
            with torch.cuda.stream(ext_stream(tr.stream() if pipelined else zk.lib.zk_ctx_stream(zk.ctx))):
                dist.all_reduce(agg, op=dist.ReduceOp.SUM)      # RCCL over xGMI: 8 x uint64 lanes
        return agg

    def barrier():
        zk.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(args.warmup):
        step(-1 - i)
    barrier()
    # keep the collector out of the timed region: a generation-2 pass over torch's object graph costs ~40 ms,
    # i.e. several whole proofs (seen as a one-off stall inside pure host code)
    gc.collect()
    gc.freeze()
    gc.disable()
    t0 = time.perf_counter()
    for i in range(args.steps):
        agg = step(i)
    t_enq = time.perf_counter() - t0          # host time to enqueue K steps (no device wait inside)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = ms_per_step / (batch * world)
    aggregate = lanes_to_fr(agg.cpu().tolist())

    # ---- per-kernel device time (HIP events on the launching stream).  Two passes over the same steps:
    #  * standalone: worker streams off, so every launch runs alone on the chip -- this is the duration a
    #    roofline can be read from (rocprofv3 of `bench.py --serial` reproduces it);
    #  * concurrent: the timed region's schedule (4 Fr groups + G1 streams overlapping), where a launch's
    #    duration includes the time it shares the chip with others.
    nprof = max(1, min(args.steps, 2))

    def profile_pass(serial):
        zk.set_flag("lanes", 0 if serial else 1)
        zk.prof_reset()
        zk.prof_enable(True)
        for i in range(nprof):
            step(10_000 + i)
        zk.sync()
        p = zk.prof_read()
        zk.prof_enable(False)
        zk.set_flag("lanes", 0 if args.serial else 1)
        return p

    prof_conc = profile_pass(serial=False) if not args.serial else {}
    prof = profile_pass(serial=True)
    total_ms = sum(v["ms"] for v in prof.values()) or 1.0
    kernels = {k: {"ms_per_step": v["ms"] / nprof, "share": v["ms"] / total_ms, "launches_per_step": v["launches"] // nprof,
                   "alg_GBps": (v["bytes"] / v["ms"] / 1e6) if v["ms"] > 0 else 0.0,
                   "concurrent_ms_per_step": (prof_conc[k]["ms"] / nprof) if k in prof_conc else None} for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
    # roofline for the dominant HBM-streaming kernel family of the timed region.  achieved = algorithmic
    # bytes per launch / average launch duration (HIP events on the launching stream).
    # family -> substrings that identify its kernel in rocprofv3's names: the batched launch form is
    # k_many<N, &body, ...>, which rocprofv3 prints mangled (_ZN2zk6k_manyILi256ETnDaXadL_ZNS_10k_sc_roundILi2ELi1EEE...)
    hbm_families = {"sumcheck_bin_round": ["k_many", "k_sc_roundILi2ELi1E"], "sumcheck_hp_round": ["k_many", "k_sc_roundILi1ELi1E"],
                    "sumcheck_ip_round": ["k_many", "k_sc_roundILi0ELi1E"], "fr_contract": ["k_many", "k_colsum"],
                    "binp_rounds": ["k_many", "k_binp_roundsILi32E"], "fr_fold": ["k_many", "k_fr_fold"]}
    cands = [k for k in kernels if k in hbm_families] or list(kernels)[:1]
    dom = max(cands, key=lambda k: kernels[k]["ms_per_step"])
    pv = prof[dom]
    achieved = pv["bytes"] / pv["ms"] / 1e6 if pv["ms"] > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        # separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (tools/pmc_summary.py);
        # gfx950 correction: FETCH_SIZE counts half the bytes of a 16-B-per-lane stream (MI355X_MICROARCH.md, HBM)
        pmc_all = json.load(open(pmc_path))
        want = hbm_families.get(dom, [])
        hits = [v for k, v in pmc_all.items() if want and all(w in k for w in want)]
        if hits:
            n = sum(h["launches"] for h in hits)
            traffic = sum((2.0 * h["fetch_kb_per_launch"] + h["write_kb_per_launch"]) * h["launches"] for h in hits) / max(n, 1) * 1024.0
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "avg_launch_ms": pv["ms"] / max(pv["launches"], 1), "alg_bytes_per_launch": pv["bytes"] / max(pv["launches"], 1),
                "launches_per_step": pv["launches"] // nprof, "measured": "standalone launches (worker streams off) after the timed region; rocprofv3 of `bench.py --serial` agrees",
                "concurrent_avg_launch_ms": (prof_conc[dom]["ms"] / max(prof_conc[dom]["launches"], 1)) if dom in prof_conc else None,
                "note": "this family is classified HBM-streaming (96*S bytes binary / 192*S ip+hadamard per table) but measures integer-VALU bound: "
                        "6-7 Fr Montgomery products (136 MACs, ~330 instructions each) per 96 B, instruction-issue saturated in the large rounds; see valu_roofline"}
    # integer-VALU view of the same kernel and of the G1 (MSM) kernels: 32x32->64 MACs per second against the
    # measured v_mad_u64_u32 issue rate (tools/instr_rate.hip: 34 T lane-MAC/s at 8 waves/SIMD)
    FR_MACS, FP_MACS = 136, 300
    macs = None
    if dom.startswith("sumcheck_"):
        per_pair = {"sumcheck_bin_round": 7, "sumcheck_hp_round": 9, "sumcheck_ip_round": 6}[dom] * FR_MACS   # Fr products per pair (6 / 8 in the factored rounds)
        bytes_per_pair = 96 if dom == "sumcheck_bin_round" else 192
        macs = pv["bytes"] / bytes_per_pair * per_pair
    valu = {"kernel": dom, "peak_TMACps": 34.0, "achieved_TMACps": (macs / pv["ms"] / 1e9) if macs else None}
    if "msm_accum" in prof:
        items = prof["msm_accum"]["bytes"] / 100.0          # ZK_LAUNCH passes 100 B per (pair, window) item
        valu["msm_accum_TMACps"] = items * 11 * FP_MACS / prof["msm_accum"]["ms"] / 1e9   # 11 Fp products per mixed add
        valu["msm_pairs_per_s_prove"] = sum(1 for _ in [0]) and (items / 26.0) / (sum(prof[k]["ms"] for k in prof if k.startswith("msm_") or k == "g1_canonicalise") / 1e3)
    roofline["valu_roofline"] = valu

    out = {"metric": "proving ms/example (batch 256) on 18M-param MLP", "value": value, "unit": "ms/example", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (Fr) / u32x12 (Fp) limbs",
           "data": "synthetic",
           "config": {"workload": args.workload, "dims": dims, "batch_per_gpu": batch, "parallelism": "independent proofs per rank, RCCL all-reduce of the aggregate",
                      "msm": args.msm},
           "ms_per_proof": ms_per_step, "host_enqueue_ms_per_step": t_enq * 1e3 / max(args.steps, 1), "aggregate_fr": hex(aggregate), "setup": wl.info, "roofline": roofline, "kernels": kernels,
           "g1_msm_mops_commit": wl.info["commit_mops"]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(zk, wl)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    zk.close()


if __name__ == "__main__":
    main()
