#!/usr/bin/env python3
"""bench.py -- proving ms/example at batch 256 on the 18M-parameter ModulusLab MLP (BASELINE.json
metric), one process per GPU, independent proof batches per rank, RCCL all-reduce of the per-proof
aggregate over xGMI.

A "step" is one pass of the reference's timed region (demo.cu:124-138): fcs[L-1].prove, then for
i = L-2..0 relus[i].prove + fcs[i].prove, over one synthetic batch that is already resident in HBM
(weights quantised + committed, inference done: all of that is setup in the reference too), plus the
transcript digest.  For N > 1 every rank proves its own batches; the digests of a rank's K proofs are summed
on the device and ONE all-reduce of that aggregate (8 x uint64 lanes, RCCL over xGMI) closes the timed region.
value = step time / (256 * N) in ms, max over ranks, exactly K steps between barriers.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mlp18m_b256|mlp18m_b1|tiny]

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child
processes, one per GPU, before this process has touched the GPU) and relays rank 0's JSON line; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

MLP18M = [784, 1000, 1773, 1773, 1773, 1773, 1773, 1124, 1000]      # model.py:14-30
WORKLOADS = {"mlp18m_b256": (MLP18M, 256), "mlp18m_b512": (MLP18M, 512), "mlp18m_b1024": (MLP18M, 1024), "mlp18m_b1": (MLP18M, 1), "tiny": ([64, 96, 48, 32], 16),
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

    def __init__(self, zk, dims, batch, seed, inflight=2, rows=None, full_tables=0):
        """rows = (lo, hi): keep only that slice of the batch (one rank's shard of ONE proof, zkdl_amd/shard.py): the
        weights, generators and the full input are the same as without it."""
        from oracle_free_gen import generator_jac
        from zkdl_amd.capi import FC, ReLU, Transcript
        self.zk, self.dims = zk, dims
        self.shapes = layer_shapes(dims)
        full_batch = batch
        if rows is not None:
            batch = rows[1] - rows[0]
        self.batch = batch
        self.B_p = 1 << ceil_log2(batch)
        rng = np.random.default_rng(1000 + seed)
        zk.seed(0xC0FFEE + seed)
        L = len(self.shapes)
        self.gens, self.fcs, self.info = [], [], {}
        t_gen = t_commit = t_full = 0.0
        full_bytes = 0
        pairs = 0
        for s in self.shapes:
            # generators = G * random scalars (demo.cu:81-82)
            t0 = time.perf_counter()
            base = zk.g1_fill(s["n_gens"], generator_jac())
            g = zk.gens(zk.g1_mul(base, zk.fr_random(s["n_gens"])))
            zk.sync()
            t_gen += time.perf_counter() - t0
            self.gens.append(g)
            if full_tables:                      # every signed digit of every window of this layer's generators: the openings' MSMs then need no buckets
                t0 = time.perf_counter()
                g.prepare_full(full_tables)
                zk.sync()
                t_full += time.perf_counter() - t0
                full_bytes += 128 * s["n_gens"] * ((254 + 2 + full_tables - 1) // full_tables) * (1 << (full_tables - 1))
            w = rng.uniform(-1.0 / np.sqrt(s["n_in"]), 1.0 / np.sqrt(s["n_in"]), size=(s["n_in"], s["n_out"])).astype(np.float32)
            wd = zk.upload_f32(w)
            zk.sync()
            t0 = time.perf_counter()
            self.fcs.append(FC.from_float(zk, s["n_in"], s["n_out"], wd, g))     # quantise + mont + commit
            zk.sync()
            t_commit += time.perf_counter() - t0
            pairs += s["in_p"] * s["out_p"]
        x = rng.standard_normal((full_batch, dims[0])).astype(np.float32)
        if rows is not None:
            x = np.ascontiguousarray(x[rows[0]:rows[1]])
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
        self.full_tables = full_tables
        self.info = dict(setup_generators_s=t_gen, setup_commit_s=t_commit, setup_full_tables_s=t_full, full_table_bytes=full_bytes, commit_pairs=pairs,
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
        # a function of the step alone (the run aggregate does not depend on how many spin-up proofs ran), in a seed domain of its own
        # (the generator randomisers draw from 0xC0FFEE + rank)
        zk.seed(0x5EEDC4A100000000 + (step & 0xffffffff))

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

VALU_PEAK_TMACPS = 34.0     # measured v_mad_u64_u32 issue peak, tools/instr_rate.hip: 33.96 T lane-MAC/s (8 waves/SIMD, every CU busy)
FR_MACS, FP_MACS = 136, 300  # 32x32->64 MACs of one Montgomery product (field.hpp: product scanning incl. the reduction)


def _rv(rng, n):
    a = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    a[:, 7] = (a[:, 7].astype(np.uint64) % 1944954707).astype(np.uint32)
    return a


def cpu_baseline(zk, wl, mode="full", parity=None, calls=None):
    """Time the oracle (CPU restatement of the reference algorithm, OpenMP over tensor elements) on the SAME workload at the SAME
    batch size, and -- `parity` = (challenges, Fr, affine G1, infinity flags) of ONE GPU proof made in the benchmarked configuration --
    compare what it returns with that proof, call by call: Fr slices bit for bit, G1 slices in affine-canonical form.
    mode "full": every call of the timed loop (demo.cu:124-138), nothing extrapolated.  mode "sample": FC[L-1] + ReLU[L-2] + FC[L-2],
    inputs = this workload's own activations / weights / generators / commitments downloaded from the device; the rest of the loop is
    extrapolated -- FC calls by their G1 scalar-mul count (m + 5 n_gens per layer, SURVEY 8d: the opening dominates and is
    batch-independent), ReLU calls by their element count.
    Returns (cpu_baseline object, parity_checked object or None)."""
    from oracle.oracle_py import Oracle
    from zkdl_amd.proof import split_calls
    orc = Oracle()
    sh = wl.shapes
    L = len(sh)
    B_p = wl.B_p
    rng = np.random.default_rng(5)
    kb = ceil_log2(B_p)
    layers = list(range(L)) if mode == "full" else [L - 1, L - 2]
    relus = list(range(L - 1)) if mode == "full" else [L - 2]
    if calls is not None:                                    # an explicit list of (kind, layer): tests
        layers = [i for kind, i in calls if kind == "fc"]
        relus = [i for kind, i in calls if kind == "relu"]
    by_call, gpu = {}, {}
    if parity is not None:
        chal, fr_g, aff_g, inf_g = parity
        by_call = {(kind, i): ch for kind, i, ch in chal}
        gpu = {(kind, i): (f, a, nf) for kind, i, f, a, nf in split_calls(wl.dims, wl.batch, fr_g, aff_g, inf_g)}
    checked = {"calls": [], "fr": 0, "g1": 0, "ok": True, "mismatch": []}

    def check(kind, i, fr_o, g1_o=None):
        if parity is None:
            return
        f, a, nf = gpu[(kind, i)]
        name = "%s%d" % ("zkFC" if kind == "fc" else "zkReLU", i)
        good = fr_o.shape == f.shape and np.array_equal(fr_o, f)
        if g1_o is not None:
            aff_o, inf_o = orc.g1_to_affine(g1_o)
            good = good and aff_o.shape == a.shape and np.array_equal(aff_o, a) and np.array_equal(inf_o, nf)
            checked["g1"] += int(a.shape[0])
        checked["fr"] += int(f.shape[0])
        checked["calls"].append(name)
        if not good:
            checked["ok"] = False
            checked["mismatch"].append(name)

    t_fc = t_relu = 0.0
    t_start = time.perf_counter()
    for i in sorted(set(layers) | set(relus), reverse=True):
        s = sh[i]
        if i not in layers:                                  # a ReLU call on its own
            Z = wl.Z[i].download()
            A_, sign, mag, rem = orc.relu_forward(Z)
            k = ceil_log2(B_p * s["out_p"])
            ch = by_call.get(("relu", i)) or [_rv(rng, n) for n in (k + 5, k + 5, k + 4, k + 4, k, k, k)]
            t0 = time.perf_counter()
            fr_o = orc.relu_prove(Z, A_, sign, mag, rem, *ch)
            t_relu += time.perf_counter() - t0
            check("relu", i, fr_o)
            del A_, sign, mag, rem
            continue
        W = wl.fcs[i].weights().download()
        gens = wl.gens[i].points().download()
        com = wl.fcs[i].com().download()
        X = (wl.A[i - 1] if i > 0 else wl.X).download()
        Z = (wl.Yhat if i == L - 1 else wl.Z[i]).download()
        ch = by_call.get(("fc", i)) or [_rv(rng, kb), _rv(rng, ceil_log2(s["in_p"])), _rv(rng, ceil_log2(s["out_p"]))]
        t0 = time.perf_counter()
        fr_o, g1_o = orc.fc_prove(X, Z, W, s["in_p"], s["out_p"], gens, com, *ch)
        t_fc += time.perf_counter() - t0
        check("fc", i, fr_o, g1_o)
        del W, gens, com, X
        if i in relus:
            A_, sign, mag, rem = orc.relu_forward(Z)            # aux tensors as the reference stores them (untimed: inference)
            k = ceil_log2(B_p * s["out_p"])
            ch = by_call.get(("relu", i)) or [_rv(rng, n) for n in (k + 5, k + 5, k + 4, k + 4, k, k, k)]
            t0 = time.perf_counter()
            fr_o = orc.relu_prove(Z, A_, sign, mag, rem, *ch)
            t_relu += time.perf_counter() - t0
            check("relu", i, fr_o)
            del A_, sign, mag, rem
    wall = time.perf_counter() - t_start
    g1_cost = lambda s_: s_["m"] + 5 * s_["n_gens"]
    fc_scale = sum(g1_cost(s_) for s_ in sh) / max(sum(g1_cost(sh[i]) for i in layers), 1)
    relu_scale = sum(B_p * sh[i]["out_p"] for i in range(L - 1)) / max(sum(B_p * sh[i]["out_p"] for i in relus), 1)
    total_s = t_fc * fc_scale + t_relu * relu_scale
    what = ("the whole prove loop (%d zkFC::prove + %d zkReLU::prove calls)" % (L, L - 1)) if mode == "full" else \
           ("zkFC::prove of layers %d,%d + zkReLU::prove of layer %d" % (L - 1, L - 2, L - 2))
    base = {"value": total_s * 1e3 / wl.batch, "unit": "ms/example (batch %d)" % wl.batch, "cores": orc.num_threads(), "kind": "port",
            "sample": "oracle (CPU restatement of the reference algorithm) on %s of this workload at batch %d: FC %.1f s + ReLU %.1f s measured "
                      "(%.0f s wall incl. downloads and the untimed aux-tensor pass); %s; the reference's CUDA binary cannot run here (no NVIDIA device)"
                      % (what, wl.batch, t_fc, t_relu, wall,
                         "nothing extrapolated" if mode == "full" else "extrapolated to all %d + %d calls: FC x%.2f by G1 scalar-mul count, ReLU x%.2f by element count" % (L, L - 1, fc_scale, relu_scale)),
            "ms_per_proof": total_s * 1e3, "measured_fc_s": t_fc, "measured_relu_s": t_relu, "extrapolated": mode != "full"}
    return base, (checked if parity is not None else None)


def cpu_baseline_check_only(zk, wl, calls, parity):
    """The oracle comparison alone, for an explicit list of (kind, layer) calls."""
    return cpu_baseline(zk, wl, "sample", parity, calls=calls)


def msm_fullwidth_probe(zk, log_pairs=24, full_tables=0):
    """G1 MSM Mop/s (BASELINE.json metric, second half): q independent MSMs over 2048 generators with FULL-WIDTH
    (255-bit, uniform) scalars, q * 2048 = 2^log_pairs (point, scalar) pairs, through zk_msm_batch; HIP-event timed."""
    from oracle_free_gen import generator_jac
    n = 2048
    q = (1 << log_pairs) // n
    g = zk.gens(zk.g1_mul(zk.g1_fill(n, generator_jac()), zk.fr_random(n)))
    if full_tables:
        g.prepare_full(full_tables)
    sc = zk.fr_random(n * q)
    out = zk.msm_batch(g, sc)                                   # builds the window tables, sizes the pools (the lane partials of 8192 MSMs are gigabytes)
    zk.sync()
    del out
    zk.timer_start()
    out = zk.msm_batch(g, sc)
    ms = zk.timer_stop_ms()
    del out, sc
    g.close()
    return {"pairs": n * q, "n_gens": n, "msms": q, "ms": ms, "mops": n * q / ms / 1e3, "scalars": "uniform 255-bit",
            "method": ("precomputed digit tables, window %d bits (no buckets)" % full_tables) if full_tables else "bucket method (Pippenger), window width by generator count (11 bits at 2048 generators)"}


def spawn_ranks(n, argv):
    """`bench.py --gpus N` outside a launcher: start the N ranks as child processes (this process never touches
    the GPU), relay rank 0's stdout, exit with the first non-zero child status."""
    import tempfile
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    rc = 0
    with tempfile.TemporaryFile() as out0:
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=out0 if r == 0 else sys.stderr))
        while any(pr.poll() is None for pr in procs):
            bad = [pr.returncode for pr in procs if pr.poll() not in (None, 0)]
            if bad:                                   # a rank died: the others would wait in the rendezvous forever
                rc = bad[0]
                for pr in procs:
                    if pr.poll() is None:
                        pr.terminate()
                break
            time.sleep(0.2)
        for pr in procs:
            try:
                pr.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pr.kill()
            if pr.returncode not in (None, 0) and rc == 0:
                rc = pr.returncode
        out0.seek(0)
        OUT.write(out0.read().decode(errors="replace"))      # rank 0's line (its own stdout is already clean of library banners)
        OUT.flush()
    return rc


def run_sharded(args, zk, dims, batch, rank, world, dist):
    """--mode shard: ONE batch-`batch` proof per step, its rows split over the ranks (strong scaling)."""
    from zkdl_amd.shard import Shard, dist_allgather, prove_sharded, prove_sharded_stream
    import torch
    per = batch // world
    ft = args.full_tables if (args.msm == "pippenger" and not args.msm_window and args.workload.startswith(("mlp18m", "tiny"))) else 0   # digit tables, as the headline
    shard = Shard(Workload(zk, dims, batch, seed=0, rows=(rank * per, (rank + 1) * per), full_tables=ft), rank, world)
    zk.set_flag("derive_rounds", 0 if args.no_derive else 1)

    def challenges(step):
        zk.seed(0xFACE + step)
        w = shard.wl
        kb = ceil_log2(batch)
        out = []
        L = len(w.shapes)
        order = [("fc", L - 1)] + [x for i in range(L - 2, -1, -1) for x in (("relu", i), ("fc", i))]
        for kind, i in order:
            s = w.shapes[i]
            if kind == "fc":
                out.append((kind, i, [zk.random_vec(kb), zk.random_vec(ceil_log2(s["in_p"])), zk.random_vec(ceil_log2(s["out_p"]))]))
            else:
                k = kb + ceil_log2(s["out_p"])
                out.append((kind, i, [zk.random_vec(n) for n in (k + 5, k + 5, k + 4, k + 4, k, k, k)]))
        return out

    gather = dist_allgather if world > 1 else (lambda obj: [obj])

    def barrier():
        zk.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # software-pipelined: five proofs in flight, every phase a sealed transcript (zkdl_amd/shard.py: prove_sharded_stream); --no-pipeline: one proof
    # at a time, three host reads each
    stream_gather = None if world == 1 else dist_allgather
    if args.no_pipeline:
        for i in range(args.warmup):
            prove_sharded([shard], challenges(-1 - i), gather)
    else:
        for _ in prove_sharded_stream(shard, [challenges(-1 - i) for i in range(args.warmup)], stream_gather):
            pass
    barrier()
    t0 = time.perf_counter()
    if args.no_pipeline:
        for i in range(args.steps):
            fr, aff, inf = prove_sharded([shard], challenges(i), gather)
    else:
        chals = (challenges(i) for i in range(args.steps))
        for fr, aff, inf in prove_sharded_stream(shard, chals, stream_gather):
            pass
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms = elapsed * 1e3 / max(args.steps, 1)
    if rank == 0:
        OUT.write(json.dumps({"metric": "proving ms/example (batch 256) on 18M-param MLP", "value": ms / batch, "unit": "ms/example", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
                          "dtype": "u32x8 (Fr) / u32x12 (Fp) limbs", "data": "synthetic",
                          "config": {"workload": args.workload, "dims": dims, "batch_total": batch, "rows_per_gpu": per,
                                     "parallelism": "one proof sharded by batch rows; two all-gathers per proof (zkdl_amd/shard.py)", "proofs_in_flight": 1 if args.no_pipeline else 5, "full_table_window_bits": ft},
                          "transcript_values": [int(fr.shape[0]), int(aff.shape[0])],
                          "transcript_sha256": __import__("hashlib").sha256(fr.tobytes() + aff.tobytes() + inf.tobytes()).hexdigest()}) + "\n")
        OUT.flush()
    if world > 1:
        dist.destroy_process_group()
    zk.close()


OUT = sys.stdout            # where the ONE line goes; main() points it at a private copy of fd 1 and sends everything else written to fd 1 to stderr
LINE_LIMIT = 4096           # the driver keeps an 8 KB tail of stdout and parses its last line: the line has to fit with room to spare (BENCH_r03 did not)


def _r(x, sig=6):
    """floats to `sig` significant digits (the line is read by people and a parser; 17-digit doubles are just bytes)"""
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def compact_line(out):
    """The ONE stdout line: the contract's keys, `roofline` + `cpu_baseline` + `parity_checked`, and the handful of scalars a reader wants
    next to the headline.  Everything else (roofline_all, per-family kernel table, PMC blocks, extensions, the latency / steady-state
    breakdowns, the G1 model, setup times) goes to the detail file."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out[k] for k in keep}
    cfg = out["config"]
    line["config"] = {k: cfg[k] for k in ("workload", "dims", "batch_per_gpu", "parallelism", "msm", "derive_rounds", "inflight", "g1_batch", "g1_streams",
                                          "full_table_window_bits", "drain_form_proofs", "prewarm_process") if k in cfg}
    r = out.get("roofline") or {}
    line["roofline"] = {k: r.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step", "alg_macs_per_launch")}
    if r.get("hbm"):
        line["roofline"]["hbm"] = {k: r["hbm"].get(k) for k in ("achieved", "peak", "unit", "frac")}
    if r.get("pmc_file"):
        line["roofline"]["traffic_source"] = r["pmc_file"]
        line["roofline"]["traffic_stale"] = r.get("traffic_stale")
    rf = out.get("roofline_chip_filling")
    if rf:
        line["roofline_chip_filling"] = {k: rf.get(k) for k in ("kernel", "frac", "avg_launch_ms", "launches_per_step")}
    c = out.get("cpu_baseline")
    if c:
        line["cpu_baseline"] = {k: c.get(k) for k in ("value", "unit", "cores", "kind", "sample", "extrapolated", "ms_per_proof")}
    p = out.get("parity_checked")
    if p:
        line["parity_checked"] = {"ok": p["ok"], "fr": p["fr"], "g1": p["g1"], "calls": len(p["calls"]), "mismatch": p["mismatch"][:4]}
    for k in ("latency_ms_per_proof", "steady_state_ms_per_proof", "g1_msm_mops", "g1_msm_mops_commit", "collective_world_size", "collective_backend",
              "per_rank_ms_per_proof", "aggregate_fr", "host_enqueue_ms_per_step"):
        if out.get(k) is not None:
            line[k] = out[k]
    bm = out.get("g1_msm_fullwidth_bucket_method")
    if bm:
        line["g1_msm_mops_bucket_method"] = bm["mops"]
    ext = out.get("extensions") or {}
    for k, short in (("sound_mode_round_by_round_fiat_shamir", "sound_mode_ms_per_proof"), ("activation_openings_in_proof", "act_openings_ms_per_proof"),
                     ("activation_commitments_and_openings_streamed", "act_commitments_and_openings_ms_per_proof")):
        if k in ext:
            line[short] = ext[k]["steady_state_ms_per_proof"]
    line["value_is"] = "pipelined throughput: wall of K proofs / K / batch; latency_ms_per_proof = ONE synchronous prove loop (demo.cu:124-140)"
    line["detail"] = out.get("detail_file")
    line = _r(line)
    for k in ("value", "ms_per_step"):          # the headline keeps its full precision (value == ms_per_step / batch / n_gpus exactly)
        line[k] = out[k]
    rl = line["roofline"]                       # ... and the fractions are those of the (rounded) numbers beside them
    if rl.get("achieved") is not None and rl.get("peak"):
        rl["frac"] = rl["achieved"] / rl["peak"]
    if rl.get("hbm") and rl["hbm"].get("achieved") is not None:
        rl["hbm"]["frac"] = rl["hbm"]["achieved"] / rl["hbm"]["peak"]
    return line


def emit(out, detail_path):
    """Write the full record to `detail_path` and print the compact line (and nothing else) on stdout."""
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
            out["detail_file"] = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
        except OSError as e:                                 # a read-only tree must not cost the headline
            sys.stderr.write("bench.py: could not write %s: %s\n" % (detail_path, e))
            out["detail_file"] = None
    line = compact_line(out)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT:                                  # never again: shed the optional keys, then the sample text
        for k in ("value_is", "per_rank_ms_per_proof", "roofline_chip_filling", "host_enqueue_ms_per_step", "aggregate_fr"):
            line.pop(k, None)
        if "cpu_baseline" in line and isinstance(line["cpu_baseline"].get("sample"), str):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:160]
        line["config"].pop("dims", None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, "bench line of %d bytes" % len(text)
    OUT.write(text + "\n")
    OUT.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="mlp18m_b256", choices=sorted(WORKLOADS))
    ap.add_argument("--msm", default="pippenger", choices=["pippenger", "literal"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", default="full", choices=["sample", "full", "off"], help="oracle timed on the whole prove loop at the bench's batch size (default: ~1 min on the "
                    "GPU box's host, nothing extrapolated), on a layer subset (extrapolated), or not at all; either way its outputs are compared with one GPU proof (parity_checked)")
    ap.add_argument("--steady-steps", type=int, default=120, help="proofs of the longer of the two untimed runs the steady-state figure is taken from (0 = skip)")
    ap.add_argument("--ext-steps", type=int, default=24, help="proofs of the longer untimed run of each protocol extension (sound mode, activation openings); 0 = skip")
    ap.add_argument("--latency-reps", type=int, default=5, help="single proofs (nothing else in flight) timed for latency_ms_per_proof (0 = skip)")
    ap.add_argument("--no-msm-probe", action="store_true", help="skip the full-width 2^24-pair zk_msm_batch throughput probe")
    ap.add_argument("--spinup-ms", type=float, default=400.0, help="untimed proofs before the W warm-up steps until this much wall time has passed: the first process on a "
                    "fresh box otherwise runs its first ~100 ms 12 %% slow (GPU power state); 0 disables")
    ap.add_argument("--prof-steps", type=int, default=3, help="proofs per event-profiled pass after the timed region (ZKDL_PROF_TRACE=<file> writes their per-launch timeline)")
    ap.add_argument("--drain", type=int, default=-1, help="the last N proofs of a run take the short G1 form (every round commitment an MSM instead of derived): "
                    "they have no later proofs to hide the longer derived form under")
    ap.add_argument("--pair", type=int, default=1, choices=[1, 2], help="proofs submitted together (zk_transcript_seal_pair): 2 = the Fr phases of two consecutive "
                    "proofs share their launches; every proof is still its own batch-256 proof with its own transcript and digest row")
    ap.add_argument("--full-tables", type=int, default=12, help="window width of the precomputed digit tables of every layer's generators (zk_gens_prepare_full: "
                    "128 B x n_gens x windows x 2^(c-1): 83 GB for the 18M MLP at c = 12); the openings' MSMs then run without buckets; 0 = the bucket method only")
    ap.add_argument("--g1-streams", type=int, default=-1, help="G1 streams sealed proofs rotate over (-1: one when --g1-batch > 1, else the library's two)")
    ap.add_argument("--g1-batch", type=int, default=3, help="sealed proofs whose G1 phases (MSM batch, derived round commitments, canonical form) run as ONE batch: "
                    "the latency-bound launches behind the accumulation are shared (zk_set_flag g1_batch); a proof then waits for its partners")
    ap.add_argument("--inflight", type=int, default=8, help="transcripts rotated by the pipelined loop (proofs whose G1 phase may still be running)")
    ap.add_argument("--no-derive", action="store_true", help="run every round commitment T_r of the openings as an MSM (library default) instead of deriving T_1.. from the "
                    "folding relation behind the MSM batch (the throughput configuration: -31 %% msm_accum work, a ~6 ms longer G1 phase per proof)")
    ap.add_argument("--serial", action="store_true", help="no worker streams at all: every launch runs alone (the schedule the roofline durations are measured under)")
    ap.add_argument("--msm-window", type=int, default=0, help="force the Pippenger window width (0 = library default)")
    ap.add_argument("--no-pipeline", action="store_true", help="finish every proof on the main stream (no overlap of proof i's G1 phase with proof i+1's Fr phase)")
    ap.add_argument("--backend", default=os.environ.get("BENCH_BACKEND", "nccl"), help="nccl (= RCCL, default) | gloo (rehearsal on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses device 0")
    ap.add_argument("--spawn", action="store_true", help="start the rank(s) as child processes even for --gpus 1 (the launcher path a multi-GPU run takes)")
    ap.add_argument("--collective", action="store_true", help="create the process group and run the aggregate's all-reduce even with one rank "
                    "(with --backend nccl: RCCL initialised and used on the single GPU a test box has)")
    ap.add_argument("--mode", default="independent", choices=["independent", "shard"],
                    help="independent (default, the bench contract): every rank proves its own batches, one all-reduce of the run aggregate; "
                         "shard: the N ranks prove ONE batch together, sharded by batch rows (zkdl_amd/shard.py: strong scaling of a single proof, "
                         "two small all-gathers per proof, synchronous orchestration -- a functional mode, not the headline)")
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "bench_detail.json"), help="where the full record goes (roofline_all, per-family kernel table, PMC blocks, extensions, "
                    "latency / steady-state breakdowns); stdout carries ONE compact JSON line (< 4 KB); empty = do not write")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the throw-away process that initialises torch on the GPU before this one does (see below)")
    ap.add_argument("--launcher-selftest", type=int, default=None, metavar="FAIL_RANK",
                    help="CPU-only check of the N-rank launch path (tests/test_dist_aggregate.py): gloo all-reduce of per-rank digests, no GPU work; "
                         "FAIL_RANK >= 0 makes that rank exit with status 3 before the rendezvous")
    args = ap.parse_args()
    # stdout carries ONE JSON line and nothing else: libraries that print there (RCCL's version banner at communicator creation, seen in the one-rank
    # rehearsal) are sent to stderr at the file-descriptor level; the line is written to a private copy of the original fd 1
    global OUT
    sys.stdout.flush()
    OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    # The first compute process on a freshly leased box runs ~12 % slow for its whole lifetime (5.2 vs 4.6 ms per proof over 20 steps;
    # any later process is not affected).  A 0.1 s helper (zkdl_amd/lib/gpu_touch: malloc, one kernel, exit) takes that role, BEFORE
    # this process -- or the ranks it spawns -- touch the GPU.  Setup work, like the commitments: not part of any timed region.
    # Measured again at the end of round 2 (tools/first_run.sh): it is the first process that initialises TORCH on the GPU that runs slow --
    # 4.5-4.6 instead of 3.95 ms per step over 20 steps, for its whole lifetime, a 6 s in-process spin-up does not help, the library's own
    # processes (pytest -m gpu, smoke) before it do not help, gpu_touch does not help -- and a throw-away process that creates one CUDA tensor
    # removes it.  So that is what runs first here (~3 s; --no-prewarm disables it and reports the cold figure).
    touch = os.path.join(ROOT, "zkdl_amd", "lib", "gpu_touch")
    if args.launcher_selftest is None:
        if "WORLD_SIZE" not in os.environ:        # this process is the only rank, or the parent of the ranks it spawns below
            devs = [str(d) for d in range(max(1, args.gpus if not args.same_device else 1))]
        else:                                     # a rank under torch.distributed.run: its own device
            devs = ["0" if args.same_device else os.environ.get("LOCAL_RANK", "0")]
        # all devices at once (N serial ~3 s helpers were N x 3 s of start-up before an 8-GPU run)
        helpers = []
        for dev in devs:
            if os.path.exists(touch) and not args.no_prewarm:   # --no-prewarm: no helper process at all (profiler passes: children would be traced too)
                helpers.append(subprocess.Popen([touch, dev]))
        for h in helpers:
            try:
                h.wait(timeout=60)
            except subprocess.TimeoutExpired:
                h.kill()
        helpers = []
        if not args.no_prewarm:
            for dev in devs:
                helpers.append(subprocess.Popen([sys.executable, "-c", "import torch; x = torch.zeros((64, 8), dtype=torch.int64, device='cuda:%s'); x.sum().item(); torch.cuda.synchronize()" % dev],
                                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
        for h in helpers:
            try:
                h.wait(timeout=300)
            except subprocess.TimeoutExpired:
                h.kill()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    if args.launcher_selftest is not None:
        from zkdl_amd.dist import allreduce_digest, fr_to_lanes
        if rank == args.launcher_selftest:
            sys.exit(3)
        dist.init_process_group("gloo")
        lanes = torch.tensor(fr_to_lanes(pow(7, rank + 1, R_MOD) * (R_MOD - 1) % R_MOD), dtype=torch.int64)
        agg = allreduce_digest(lanes)
        dist.barrier()
        seen = [None] * world
        dist.all_gather_object(seen, rank)
        if rank == 0:
            OUT.write(json.dumps({"selftest": True, "n_gpus": world, "collective_world_size": dist.get_world_size(), "aggregate_fr": hex(agg), "ranks": sorted(seen)}) + "\n")
            OUT.flush()
        dist.destroy_process_group()
        return
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll = world > 1 or args.collective           # a process group exists (N > 1, or one rank with --collective)
    if coll:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    from zkdl_amd.capi import MSM_LITERAL, MSM_PIPPENGER, Zk
    zk = Zk(local_rank)
    zk.set_msm_mode(MSM_PIPPENGER if args.msm == "pippenger" else MSM_LITERAL)
    if args.msm_window:
        zk.set_msm_window(args.msm_window)
    derive = not args.no_derive and args.msm == "pippenger"
    zk.set_flag("derive_rounds", 1 if derive else 0)
    zk.set_flag("g1_batch", max(1, args.g1_batch))
    g1_streams = args.g1_streams if args.g1_streams >= 0 else (1 if args.g1_batch > 1 else 0)
    zk.set_flag("g1_streams", g1_streams)
    for kv in os.environ.get("ZKDL_BENCH_FLAGS", "").split(","):       # e.g. ZKDL_BENCH_FLAGS=derive_rounds=1,relu_bytes=0 (A/B runs)
        if "=" in kv:
            zk.set_flag(kv.split("=")[0], int(kv.split("=")[1]))
    dims, batch = WORKLOADS[args.workload]
    if args.mode == "shard":
        return run_sharded(args, zk, dims, batch, rank, world, dist)
    ft = args.full_tables if (args.msm == "pippenger" and not args.msm_window and args.workload.startswith(("mlp18m", "tiny"))) else 0
    wl = Workload(zk, dims, batch, seed=rank, inflight=args.inflight, full_tables=ft)

    # The G1 phase of every proof (MSM batch, canonicalisation, digest) is ordered on the library's G1 stream; the Fr
    # phase of the next proof runs beside it on the worker lanes.  Every proof leaves its digest (8 x uint64 lanes)
    # in its own row of `rows`; no torch op and no collective runs per proof.  After the K-th proof the rows are
    # summed on the device and the rank aggregate is all-reduced ONCE (north_star: "an RCCL all-reduce over xGMI only
    # for the final sumcheck aggregate") -- ranks never couple their step times.
    pipelined = not args.no_pipeline and not args.serial
    if args.serial:
        zk.set_flag("lanes", 0)
    nprof = max(1, min(args.steps, args.prof_steps))
    rows = torch.zeros((args.steps + 1, 8), dtype=torch.int64, device="cuda")
    scratch_rows = torch.zeros((max(args.warmup, 1) + 4 * nprof + 8 + 16, 8), dtype=torch.int64, device="cuda")
    extra_rows = scratch_rows.data_ptr() + 64 * (max(args.warmup, 1) + 4 * nprof + 8)     # 16 rows for the untimed side measurements below
    rows_base, scratch_base = rows.data_ptr(), scratch_rows.data_ptr()
    torch.cuda.synchronize()

    def step(i, row_ptr):
        chal = wl.challenges(i)
        tr = wl.trs[i % len(wl.trs)] if pipelined else wl.tr
        wl.prove(chal, tr)
        if pipelined:
            tr.seal(row_ptr)
        else:
            tr.digest_dev(row_ptr)

    paired = pipelined and args.pair == 2 and len(wl.trs) >= 2

    def step2(i, row_ptr, j, row_ptr_j):
        """proofs i and j recorded on two transcripts, then submitted together"""
        ta, tb = wl.trs[i % len(wl.trs)], wl.trs[j % len(wl.trs)]
        wl.prove(wl.challenges(i), ta)
        wl.prove(wl.challenges(j), tb)
        ta.seal_pair(tb, row_ptr, row_ptr_j)

    def barrier():
        zk.sync()
        torch.cuda.synchronize()
        if coll:
            dist.barrier()

    t_spin = time.perf_counter()
    n_spin = 0
    # ... and at least two rounds over the transcripts in flight: the pools must have seen every in-flight shape before the timed region (deep4096x64: a proof's
    # bucket arrays are gigabytes, and a run that still grows the pools spends 100-400 ms per step in allocation calls on the host)
    min_spin = 2 * len(wl.trs) if (pipelined and args.spinup_ms > 0) else 0
    while (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms or n_spin < min_spin:
        step(-100000 - n_spin, scratch_base)
        n_spin += 1
        if n_spin % 8 == 0:
            zk.sync()                            # do not let the host run ahead of the device during the spin-up
    zk.sync()
    i = 0
    while i < args.warmup:
        # the warm-up also runs proofs in the drain configuration (see below), so that the allocator pools hold blocks of both shapes
        if derive and args.warmup >= 2 and i >= args.warmup - 2:
            zk.lib.zk_set_flag(zk.ctx, b"derive_rounds", 0)
        if paired and i + 1 < args.warmup:
            step2(-1 - i, scratch_base + 64 * i, -2 - i, scratch_base + 64 * (i + 1))
            i += 2
        else:
            step(-1 - i, scratch_base + 64 * i)
            i += 1
        if derive:
            zk.lib.zk_set_flag(zk.ctx, b"derive_rounds", 1)
    # warm what closes the timed region as well: the row sum is a torch kernel whose first launch in a process loads its code object
    # (~7 ms measured: 0.35 ms per step of a 20-step run; tools/notorch_bench.py runs the same loop without torch: 3.52 ms per proof at K = 20 and K = 100)
    warm_agg = rows[:args.steps].sum(dim=0)
    if coll:                                # ... and the communicator
        dist.all_reduce(warm_agg, op=dist.ReduceOp.SUM)
    barrier()
    # keep the collector out of the timed region: a generation-2 pass over torch's object graph costs ~40 ms,
    # i.e. several whole proofs (seen as a one-off stall inside pure host code)
    gc.collect()
    gc.freeze()
    gc.disable()
    # Derived round commitments trade a longer G1 phase per proof (~11 ms instead of ~5) for less vector work: worth it while there
    # are later proofs to hide it under.  The LAST proofs of the run have none, so they take the short G1 phase (every T_r as an MSM):
    # the pipeline then drains in one MSM chain instead of two.
    if args.drain < 0:
        args.drain = max(1, args.g1_batch)        # the whole last G1 batch: one derived proof in it would keep the derivation on the batch's chain
    drain_from = args.steps - args.drain if derive else args.steps
    t0 = time.perf_counter()
    i = 0
    while i < args.steps:
        if derive and i == drain_from:
            # the run's last proofs have no later proof to hide a long G1 chain under: short form (every round commitment an MSM), and each one's
            # G1 phase on its own as soon as its Fr phase is submitted, so that only the very last proof's G1 chain is left when the Fr work ends
            zk.lib.zk_set_flag(zk.ctx, b"derive_rounds", 0)
            zk.lib.zk_set_flag(zk.ctx, b"g1_batch", 1)
        if paired and i + 1 < args.steps:
            step2(i, rows_base + 64 * i, i + 1, rows_base + 64 * (i + 1))
            i += 2
        else:
            step(i, rows_base + 64 * i)
            i += 1
    t_enq = time.perf_counter() - t0          # host time to enqueue K steps (no device wait inside)
    zk.sync()                                 # all G1 streams joined: every row is final
    agg = rows[:args.steps].sum(dim=0)        # < K * 2^32 per lane: exact in int64
    if coll:
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)      # RCCL over xGMI: 8 x uint64 lanes, once per run
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    zk.set_flag("derive_rounds", 1 if derive else 0)
    zk.set_flag("g1_batch", max(1, args.g1_batch))
    per_rank_ms = [elapsed * 1e3 / max(args.steps, 1)]
    if coll:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [float(x.item()) * 1e3 / max(args.steps, 1) for x in allt]
        elapsed = max(float(x.item()) for x in allt)
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = ms_per_step / (batch * world)
    rank_devices = [local_rank]
    if coll:                                        # which device every rank proved on, as the ranks themselves report it
        dv = torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device="cuda")
        alld = [torch.zeros_like(dv) for _ in range(world)]
        dist.all_gather(alld, dv)
        rank_devices = [int(x.item()) for x in alld]
    aggregate = lanes_to_fr(agg.cpu().tolist())
    if os.environ.get("ZKDL_BENCH_DUMP_ROWS"):       # per-proof digests, for run-to-run comparisons
        np.save(os.environ["ZKDL_BENCH_DUMP_ROWS"], rows[:args.steps].cpu().numpy())

    # ---- side measurements, all OUTSIDE the timed region (same process, same workload):
    #  * steady_state_ms_per_proof: two untimed runs of N/3 and N proofs; the difference of their wall times over the difference of their
    #    lengths cancels the pipeline's fill and drain, which the K-step headline pays (ms_per_step above).
    #  * latency_ms_per_proof: ONE proof with nothing else in flight, host-timed from the first prove call to the end of its G1 phase -- the
    #    reference's own definition of its metric (demo.cu:124-140 times one prove loop); both forms of the opening's round commitments.
    gc.disable()

    def run_n(n, base_step):
        zk.sync()
        t = time.perf_counter()
        for j in range(n):
            step(base_step + j, extra_rows + 64 * (j % len(wl.trs)))
        zk.sync()
        return (time.perf_counter() - t) * 1e3

    steady = None
    if args.steady_steps >= 12 and pipelined and not paired:
        n_long, n_short = args.steady_steps, max(4, args.steady_steps // 3)
        t_short = run_n(n_short, 40_000)
        t_long = run_n(n_long, 50_000)
        steady = {"ms_per_proof": (t_long - t_short) / (n_long - n_short), "runs": {str(n_short): t_short, str(n_long): t_long},
                  "how": "(wall of %d proofs - wall of %d proofs) / %d, both untimed runs after the timed region, derive_rounds as benchmarked, no drain form" % (n_long, n_short, n_long - n_short)}
    latency = None
    if args.latency_reps > 0:
        latency = {}
        # a single proof's G1 phase runs alone ("g1_batch" = 1).  It still shares the chip with its OWN Fr phase (it starts behind the openings'
        # scalar preparation), and measures faster on the half-chip G1 stream than on one over all compute units (5.1 against 5.5 ms): kept as
        # "msm_rounds_full_chip_g1_stream" for the record
        zk.set_flag("g1_batch", 1)
        for name, flag, mask in (("msm_rounds", 0, 1), ("derive_rounds", 1, 1), ("msm_rounds_full_chip_g1_stream", 0, 0)):
            if flag and not derive:
                continue
            zk.set_flag("derive_rounds", flag)
            zk.set_flag("g1_mask", mask)
            ts = [run_n(1, 60_000 + 100 * flag + 1000 * mask + j) for j in range(args.latency_reps + 1)][1:]
            latency[name] = {"median_ms": float(np.median(ts)), "min_ms": float(min(ts)), "reps": len(ts)}
        zk.set_flag("derive_rounds", 1 if derive else 0)
        zk.set_flag("g1_mask", 1)
        zk.set_flag("g1_batch", max(1, args.g1_batch))
        latency["definition"] = "one batch-%d proof, nothing else in flight: host wall time from the first prove call until its G1 phase has finished (demo.cu:124-140)" % batch
    # ---- the two protocol extensions, same pipelined loop (untimed side measurements; --ext-steps 0 skips them):
    #  * sound mode: every sumcheck round challenge from a hash of the rounds before it (zk_fc_prove_fs / zk_relu_prove_fs, csrc/fs_rounds.hip)
    #  * activation openings inside the proof: per hidden layer one claim-reduction sumcheck + one opening of [Z_i ; A_i] (zk_act_open); the
    #    14 activation commitments themselves are made once, outside (they are per-input setup, like inference)
    extensions = None
    if args.ext_steps > 0 and pipelined and not paired and args.msm == "pippenger":
        from zkdl_amd.proof import (act_claim_points, act_open_challenges, fs_challenges, prove_mlp_fs, statement_digest)
        coms_ = [fc.com() for fc in wl.fcs]
        seed_ = statement_digest(zk, dims, batch, wl.gens, coms_, wl.X, wl.Yhat)
        chal_ = fs_challenges(seed_, dims, batch)
        pts_, ac_ = act_claim_points(dims, chal_), act_open_challenges(seed_, dims, batch)   # timing only: no activation commitments are made here, so the seed does not absorb them

        def run_ext(n, which):
            zk.sync()
            t = time.perf_counter()
            for j in range(n):
                tr_ = wl.trs[j % len(wl.trs)]
                if which == "fs":
                    prove_mlp_fs(wl.fcs, wl.relus, wl.gens, wl.X, wl.Z, wl.A, wl.Yhat, seed_, tr_)
                else:
                    wl.prove(chal_, tr_)
                    for i_, (alpha_, fold_) in enumerate(ac_):
                        zk.act_open(wl.gens[i_], wl.Z[i_], wl.A[i_], pts_[i_], alpha_, fold_, tr_)
                tr_.seal(extra_rows + 64 * (j % len(wl.trs)))
            zk.sync()
            return (time.perf_counter() - t) * 1e3

        extensions = {}
        for which in ("fs", "act"):
            run_ext(len(wl.trs), which)                              # warm: bit tensors materialised, pools sized
            n_short, n_long = max(2, args.ext_steps // 3), args.ext_steps
            t_short, t_long = run_ext(n_short, which), run_ext(n_long, which)
            one = [run_ext(1, which) for _ in range(3)]
            extensions["sound_mode_round_by_round_fiat_shamir" if which == "fs" else "activation_openings_in_proof"] = {
                "steady_state_ms_per_proof": (t_long - t_short) / (n_long - n_short), "latency_ms_per_proof": float(np.median(one)),
                "runs_ms": {str(n_short): t_short, str(n_long): t_long}}
        # the activation commitments themselves as part of the pipeline (zk_act_commit: deferred row MSMs on the digit tables, two sealed stages per proof,
        # three proofs in flight; zkdl_amd/proof.py: prove_with_activation_commitments_stream)
        from zkdl_amd.proof import prove_with_activation_commitments_stream

        def run_actc(n):
            zk.sync()
            t = time.perf_counter()
            for _ in prove_with_activation_commitments_stream(zk, dims, batch, wl.gens, wl.fcs, wl.relus, wl.X, wl.Z, wl.A, wl.Yhat, seed_, n):
                pass
            zk.sync()
            return (time.perf_counter() - t) * 1e3

        run_actc(3)
        n_short, n_long = max(2, args.ext_steps // 4), max(4, args.ext_steps // 2)
        t_short, t_long = run_actc(n_short), run_actc(n_long)
        extensions["activation_commitments_and_openings_streamed"] = {"steady_state_ms_per_proof": (t_long - t_short) / (n_long - n_short),
                                                                     "latency_ms_per_proof": run_actc(1), "runs_ms": {str(n_short): t_short, str(n_long): t_long},
                                                                     "what": "per proof: 14 row-wise commitments of the 2^18 .. 2^19-entry activation tensors + the layer proofs + 7 reductions and openings"}
        extensions["note"] = ("same workload, same pipelined loop as the headline (derive_rounds / g1_batch / inflight as configured); the headline itself is the "
                              "reference-parity mode (pre-drawn challenges, no activation commitments), which is what the reference times")
    # ---- the proof the CPU oracle is compared with: made in the benchmarked configuration, with len(trs) proofs in flight around it
    parity_src = None
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline and args.cpu_baseline != "off"
    if want_cpu and args.msm == "pippenger":
        P = 70_000
        nfl = len(wl.trs) if pipelined else 1
        for j in range(nfl):
            step(P - (nfl - 1) + j, extra_rows + 64 * (8 + j % 8))
        trp = wl.trs[P % len(wl.trs)] if pipelined else wl.tr
        fr_p, _ = trp.read()
        aff_p, inf_p = trp.read_affine()
        parity_src = (wl.challenges(P), fr_p, aff_p, inf_p)
    gc.enable()

    # ---- per-kernel device time (HIP events on the launching stream).  Two passes over the same steps:
    #  * standalone: worker streams off, so every launch runs alone on the chip -- this is the duration a
    #    roofline can be read from (rocprofv3 of `bench.py --serial` reproduces it);
    #  * concurrent: the timed region's schedule (4 Fr groups + G1 streams overlapping), where a launch's
    #    duration includes the time it shares the chip with others.
    prof_row = [0]

    def profile_pass(serial):
        zk.set_flag("lanes", 0 if serial else 1)
        zk.prof_reset()
        zk.prof_enable(True)
        for i in range(nprof):
            step(10_000 + i, scratch_base + 64 * (max(args.warmup, 1) + prof_row[0]))
            prof_row[0] += 1
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
    # ---- rooflines.  Every family on this path is wide-integer modular arithmetic and measures integer-VALU bound (profiles/: SQ
    # counters), so the bound is "valu":
    #   achieved = ALGORITHMIC 32x32->64 MACs per launch (Montgomery products the kernel's algorithm needs x MACs per product in the
    #              problem's own 32-bit words: 136 for Fr, 300 for Fp) / average launch duration (HIP events on the
    #              launching stream, standalone launches)
    #   peak     = measured v_mad_u64_u32 issue peak (tools/instr_rate.hip)
    # `roofline_all` lists every family that has a model; `roofline` is the family with the most device time per proof among them
    # (whatever its occupancy), `roofline_chip_filling` the longest one whose launches fill the chip.  The HBM view of the same kernel
    # (algorithmic bytes / duration vs 8 TB/s, PMC traffic) is kept under "hbm".
    # family -> (substrings identifying its kernel in rocprofv3's mangled names, algorithmic bytes per work unit, Fr products per unit):
    #   binary round: unit = a pair, 96 B (two reads + one write), 6 products (a0 w, da w, three coefficients, the fold; 7 when the weight does not factor)
    #   Hadamard / inner-product round: unit = a pair of both tables, 192 B, 7 / 5 products ; contraction: unit = an element, 32 B, 1 product
    fam = {"sumcheck_bin_round": (["k_many", "k_sc_round29ILi2E"], 96, 6), "sumcheck_hp_round": (["k_many", "k_sc_round29ILi1E"], 192, 7),
           "sumcheck_ip_round": (["k_many", "k_sc_roundILi0ELi1E"], 192, 5), "fr_contract": (["k_many", "k_colsum"], 32, 1),
           # byte-histogram pass: unit = one 32-bit magnitude word + one 16-bit remainder word (4 + 32 and 2 + 32 bytes), 7 + 3 products
           # (word weights for the sumcheck and the partial_me, the cross terms, the round-4 product, the folded output)
           "binp_words": (["k_many", "k_binb_words"], 70, 10), "msm_accum": (["k_msm_accum"], 100, None), "msm_combine": (["k_msm_combine"], None, None),
           "msm_bucket_reduce": (["k_msm_rowcol"], None, None), "msm_horner": (["k_msm_weighted"], None, None), "msm_derive": (["k_derive_mul"], None, None),
           "g1_canonicalise": (["k_g1_canonicalise"], None, None)}
    macs_per_step = {}
    for k, v in prof.items():
        if k in fam and fam[k][2] is not None:
            macs_per_step[k] = v["bytes"] / nprof / fam[k][1] * fam[k][2] * FR_MACS
    # MSM items of one proof: the launch is sized for n * W slots per job (its `bytes`), but T0_r / T1_r only touch every other
    # generator (their other scalars are zero and are dropped by the sort): per layer 2 k_lo n + n (T, T0, T1 of every round, the
    # final generator) + m (com(u_hi)) nonzero full-width scalars, 26 ten-bit windows each
    # (with derived round commitments all k full-size T_r MSMs are gone: (k + 1) n + m)
    klo = [ceil_log2(s_["n_gens"]) for s_ in wl.shapes]
    real_pairs = sum(((k_ + 1) if derive else (2 * k_ + 1)) * s_["n_gens"] + s_["m"] for k_, s_ in zip(klo, wl.shapes))
    g1_model = None
    ft_on = bool(getattr(wl, "full_tables", 0)) and "msm_table_accum" in prof
    if ft_on:
        # the openings' MSMs on precomputed digit tables: (real_pairs - the com(u_hi) MSMs over the commitment points) x windows mixed additions, then one
        # addition per lane partial; only the 8 com(u_hi) MSMs (m points each) are left on the bucket path below
        W_ft = (254 + 2 + wl.full_tables - 1) // wl.full_tables
        ft_pairs = real_pairs - sum(s_["m"] for s_ in wl.shapes)
        macs_per_step["msm_table_accum"] = ft_pairs * W_ft * 10 * FP_MACS
        macs_per_step["msm_table_sum"] = ft_pairs * W_ft / 16.0 * 14 * FP_MACS      # one partial per 16 items (msm.hip: MSM_CHUNK x FT_SUBS)
        fam["msm_table_accum"] = (["k_msmf_accum"], 100, None)
        fam["msm_table_sum"] = (["k_msmf_reduce"], None, None)
    if args.msm == "pippenger":
        c_win, chunk = (args.msm_window or 10), 16
        W_win = (254 + 2 + c_win - 1) // c_win
        has_bucket = "msm_accum" in prof
        slots = (prof["msm_accum"]["bytes"] / 100.0 / nprof) if has_bucket else 0.0
        bucket_pairs = 0 if (ft_on and not has_bucket) else (sum(s_["m"] for s_ in wl.shapes) if ft_on else real_pairs)
        items = min(slots, bucket_pairs * float(W_win))
        n_jobs = 0 if not has_bucket else (len(klo) if ft_on else sum((2 * k_ + 2) if derive else (3 * k_ + 2) for k_ in klo))   # MSMs on the bucket path
        B = 1 << (c_win - 1)
        LB = c_win // 2
        nlo, nhi = 1 << LB, (B >> LB) + 1
        ADD, MADD, DBL = 14, 10, 9                                                    # XYZZ: 12M+2S, 8M+2S, 6M+3S (msm.hip)
        JADD, JDBL, INV = 16, 7, 485                                                  # Jacobian 11M+5S, 2M+5S (g1q.hpp); Fermat inverse with 4-bit windows
        n_derived = sum(k_ * k_ for k_ in klo) if derive else 0                       # coef * point products behind the batch (T_0 .. T_{k-1}: 2r + 1 each)
        n_g1_out = sum(3 * k_ + 2 for k_ in klo)                                      # transcript points put into canonical form
        g1_model = {"bucket_items": items, "bucket_jobs": n_jobs, "derived_products": n_derived, "points_canonicalised": n_g1_out, "window_bits": c_win, "windows": W_win,
                    "table_items": (ft_pairs * W_ft if ft_on else 0), "table_window_bits": (wl.full_tables if ft_on else None), "real_pairs": real_pairs}
        if has_bucket:
            macs_per_step["msm_accum"] = items * MADD * FP_MACS
            # stitching: a bucket that spans chunks t0..t1 costs t1 - t0 additions: in total one per chunk boundary that falls inside a bucket (~ all of them)
            macs_per_step["msm_combine"] = max(items / chunk - n_jobs, 0.0) * ADD * FP_MACS
            macs_per_step["msm_bucket_reduce"] = 2.0 * B * n_jobs * ADD * FP_MACS     # every bucket enters one row sum and one column sum
            # per MSM: c bit-slices of the row / column sums (half of the sums each), sum_j j doublings, a 15-addition tree, back to Jacobian (6 products)
            macs_per_step["msm_horner"] = n_jobs * ((LB * (nlo // 2) + (c_win - LB) * ((nhi - 1) // 2) + 15) * ADD + (c_win * (c_win - 1) // 2) * DBL + 6) * FP_MACS
        if n_derived and "msm_derive" in prof:
            # per product: GLV halves on two lanes, each 128 doublings + 32 window additions + an 8-entry table (1 doubling + 6 additions); the halves' sum;
            # then one addition per product into its T_r
            macs_per_step["msm_derive"] = n_derived * (2 * (128 * JDBL + 32 * JADD + JDBL + 6 * JADD) + JADD + JADD) * FP_MACS
        if "g1_canonicalise" in prof:
            # 1/Z by binary GCD (fpinv.hpp: 25 groups x ten 12-limb multiply-accumulate rows, no field products) + the two domain conversions, the R^3
            # product, Z^-2, Z^-3, x, y and their conversions back
            macs_per_step["g1_canonicalise"] = n_g1_out * (25 * 10 * 12 + 10 * FP_MACS)
    # the newest counter summary under profiles/ (tools/collect_profiles.sh + tools/pmc_summary.py: separate rocprofv3 --pmc passes of this bench).
    # It records the fingerprint of the sources it was collected from; if this tree differs, the traffic figure is flagged stale.
    import glob as _glob
    from src_fingerprint import source_fingerprint
    pmc_files = sorted(_glob.glob(os.path.join(ROOT, "profiles", "pmc_r[0-9][0-9].json")))
    pmc_path = pmc_files[-1] if pmc_files else None
    pmc_all = json.load(open(pmc_path)) if pmc_path else {}
    pmc_name = ("profiles/" + os.path.basename(pmc_path)) if pmc_path else None
    pmc_stale = bool(pmc_all) and pmc_all.get("src_sha256") != source_fingerprint()

    def pmc_of(family):
        want = fam[family][0] if family in fam else [family]
        hit = None
        for name, rec in pmc_all.get("kernels", {}).items():
            if all(w in name for w in want):
                hit = rec if hit is None or rec.get("launches", 0) > hit.get("launches", 0) else hit
        return hit

    # Device time per proof in the TIMED configuration.  The standalone pass runs every proof's G1 phase by itself; the timed region runs g1_batch
    # proofs' phases as one batch, where the launches that are one dependent chain on a handful of waves (derived round commitments, canonical form)
    # take as long for three proofs as for one.  Their launch count per proof is taken from the concurrent pass (the timed region's schedule, nprof a
    # multiple of g1_batch) and priced at the standalone launch duration; every other family does work proportional to the proofs in it.
    shared_chain = ("msm_derive", "g1_canonicalise")

    def timed_launches(k):
        return (prof_conc[k]["launches"] / nprof) if k in prof_conc else prof[k]["launches"] / nprof

    def timed_ms(k):
        per_step = prof[k]["ms"] / nprof
        if k in shared_chain and k in prof_conc and prof[k]["launches"]:
            return per_step * min(1.0, prof_conc[k]["launches"] / prof[k]["launches"])
        return per_step

    def roofline_of(k):
        pv = prof[k]
        launches = max(pv["launches"], 1)
        avg_ms = pv["ms"] / launches
        macs_launch = macs_per_step[k] * nprof / launches
        ach = macs_launch / avg_ms / 1e9 if avg_ms > 0 else 0.0
        hbm_ach = pv["bytes"] / pv["ms"] / 1e6 if pv["ms"] > 0 else 0.0
        hit = pmc_of(k)
        traffic = (2.0 * hit["fetch_kb_per_launch"] + hit["write_kb_per_launch"]) * 1024.0 if hit and "fetch_kb_per_launch" in hit else None
        return {"kernel": k, "bound": "valu", "achieved": ach, "peak": VALU_PEAK_TMACPS, "unit": "TMAC/s (32x32->64 integer multiply-accumulates per second)",
                "frac": ach / VALU_PEAK_TMACPS, "traffic": traffic, "avg_launch_ms": avg_ms, "alg_macs_per_launch": macs_launch, "ms_per_step": pv["ms"] / nprof,
                "launches_per_step": pv["launches"] // nprof,
                "timed_region_launches_per_step": timed_launches(k), "timed_region_ms_per_step": timed_ms(k),
                "concurrent_avg_launch_ms": (prof_conc[k]["ms"] / max(prof_conc[k]["launches"], 1)) if k in prof_conc else None,
                "hbm": {"achieved": hbm_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_ach / HBM_PEAK_GBPS, "alg_bytes_per_launch": pv["bytes"] / launches,
                        "traffic_bytes_per_launch": traffic},
                "pmc": ({kk: hit[kk] for kk in hit if kk not in ("fetch_kb_per_launch", "write_kb_per_launch")} if hit else None),
                "pmc_source": (("%s, collected at commit %s: a separate rocprofv3 --pmc run of this bench, not this run" % (pmc_name, pmc_all.get("commit", "?"))) if hit else None),
                "pmc_file": (pmc_name if hit else None), "traffic_stale": (pmc_stale if hit else None)}

    modelled = [k for k in kernels if k in macs_per_step and prof[k]["ms"] > 0]
    roofline_all = [roofline_of(k) for k in modelled]
    for r_ in roofline_all:
        kernels[r_["kernel"]]["alg_TMACps"] = r_["achieved"]
        kernels[r_["kernel"]]["valu_frac"] = r_["frac"]
    if modelled:
        dom = max(modelled, key=timed_ms)
        roofline = dict(roofline_of(dom))
        # chip-filling: launches of at least ~4 waves per SIMD's worth of workgroups; on this path that is the accumulation and the big rounds
        filling = [k for k in modelled if k in ("msm_table_accum", "msm_accum", "sumcheck_bin_round", "sumcheck_hp_round", "binp_words", "fr_contract")]
        fill = max(filling, key=lambda k: kernels[k]["ms_per_step"]) if filling else dom
        roofline["measured"] = "standalone launches (worker streams off) after the timed region, HIP events on the launching stream; rocprofv3 of `bench.py --serial` agrees (profiles/)"
        roofline["selection"] = ("the family with the most device time per proof IN THE TIMED CONFIGURATION (timed_region_ms_per_step: standalone launch duration x launches per proof "
                                 "of the timed schedule, where g1_batch proofs share the launches of the derived-commitment and canonical-form chains) among all families with a "
                                 "MAC model (roofline_all lists every one, with ms_per_step of an unbatched proof beside it); roofline_chip_filling = the longest family whose launches fill the chip")
        roofline["note"] = ("wide-integer modular arithmetic: no MFMA mapping; HBM traffic (PMC) is at or below the algorithmic bytes (window tables hit L2), the SIMDs are "
                            "instruction-issue bound; peak = measured issue rate of the MAC instruction, so frac is the share of issue slots doing algorithmic MACs. "
                            "Families that run on a handful of waves (msm_derive, g1_canonicalise, msm_horner: dependent chains of field products) are latency-bound: "
                            "their frac is the price of the chain, not of the instruction mix")
        roofline_fill = roofline_of(fill)
    else:
        k0 = next(iter(kernels))
        roofline = {"kernel": k0, "bound": "valu", "achieved": None, "peak": VALU_PEAK_TMACPS, "unit": "TMAC/s", "frac": None, "traffic": None,
                    "avg_launch_ms": prof[k0]["ms"] / max(prof[k0]["launches"], 1), "note": "no MAC model for this configuration"}
        roofline_fill = None

    out = {"metric": "proving ms/example (batch 256) on 18M-param MLP", "value": value, "unit": "ms/example", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (Fr) / u32x12 (Fp) limbs",
           "data": "synthetic",
           "config": {"workload": args.workload, "dims": dims, "batch_per_gpu": batch, "parallelism": "independent proofs per rank, one RCCL all-reduce of the run aggregate",
                      "msm": args.msm, "derive_rounds": bool(derive), "inflight": args.inflight, "g1_batch": args.g1_batch, "g1_streams": (g1_streams or 2), "full_table_window_bits": ft, "proofs_per_submission": 2 if paired else 1, "prewarm_process": not args.no_prewarm,
                      "drain_form_proofs": (min(args.drain, args.steps) if derive else 0), "spinup_ms": args.spinup_ms, "spinup_proofs": n_spin,
                      "warmup_drain_form_proofs": (min(2, args.warmup) if derive and args.warmup >= 2 else 0),
                      "schedule_note": "the last drain_form_proofs proofs of the timed region run every round commitment as an MSM (derive_rounds off) and their G1 phases unbatched (g1_batch 1): they have no later proof to hide a long G1 chain under"},
           "ms_per_proof": ms_per_step, "steady_state_ms_per_proof": (steady["ms_per_proof"] if steady else None), "steady_state": steady,
           "latency_ms_per_proof": (min(v["median_ms"] for v in latency.values() if isinstance(v, dict)) if latency else None), "latency": latency, "extensions": extensions,
           "per_rank_ms_per_proof": per_rank_ms, "collective_world_size": (dist.get_world_size() if coll else 1),
           "collective_backend": (dist.get_backend() if coll else None), "rank_devices": rank_devices,
           "host_enqueue_ms_per_step": t_enq * 1e3 / max(args.steps, 1), "aggregate_fr": hex(aggregate), "setup": wl.info, "roofline": roofline,
           "roofline_chip_filling": roofline_fill, "roofline_all": roofline_all, "g1_model": g1_model, "kernels": kernels,
           "g1_msm_mops_commit": wl.info["commit_mops"]}
    if "msm_accum" in prof or "msm_table_accum" in prof:
        g1_ms = sum(prof[k]["ms"] for k in prof if (k.startswith("msm_") and k != "msm_fulltable_build") or k == "g1_canonicalise") / nprof
        out["g1_msm_mops_prove"] = real_pairs / g1_ms / 1e3 if g1_ms > 0 else None   # nonzero full-width (point, scalar) pairs of one proof / its G1 kernels' time
    if not args.no_msm_probe and args.workload.startswith("mlp18m"):
        # BASELINE's second metric, "G1 MSM Mop/s at 1/2/4/8 GPUs": every rank runs the same 2^24-pair probe on its own device at the same time
        # (independent MSMs over replicated generators: the rows-of-W partitioning of SURVEY 8e, no exchange); the job figure is all ranks' pairs
        # over the slowest rank's time
        if coll:
            dist.barrier()
        mine = msm_fullwidth_probe(zk, full_tables=wl.full_tables)
        out["g1_msm_fullwidth"] = mine
        if coll:
            t = torch.tensor([mine["ms"]], dtype=torch.float64, device="cuda")
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            ms_all = [float(x.item()) for x in allt]
            out["g1_msm_fullwidth"] = dict(mine, per_rank_ms=ms_all, ranks=world, pairs_all_ranks=mine["pairs"] * world, mops_all_ranks=mine["pairs"] * world / max(ms_all) / 1e3)
            out["g1_msm_mops"] = out["g1_msm_fullwidth"]["mops_all_ranks"]
        else:
            out["g1_msm_mops"] = mine["mops"]
        if wl.full_tables and rank == 0 and world == 1:
            out["g1_msm_fullwidth_bucket_method"] = msm_fullwidth_probe(zk, full_tables=0)
    if coll:
        dist.barrier()
    rc = 0
    if want_cpu:
        out["cpu_baseline"], checked = cpu_baseline(zk, wl, args.cpu_baseline, parity_src)
        if checked is not None:
            checked["gpu_configuration"] = "derive_rounds=%d, %d proofs in flight, sealed on the G1 streams (the timed region's configuration)" % (1 if derive else 0, len(wl.trs) if pipelined else 1)
            out["parity_checked"] = checked
            if not checked["ok"]:
                rc = 4
    if rank == 0:
        emit(out, args.detail_out)
    if coll:
        dist.destroy_process_group()
    zk.close()
    if rc:
        sys.stderr.write("bench.py: the GPU transcript differs from the oracle's (parity_checked.mismatch)\n")
        sys.exit(rc)


if __name__ == "__main__":
    main()
