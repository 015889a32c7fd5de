"""ONE proof sharded over N GPUs by batch (the reference lists multi-GPU as future work).

Every activation-side table of the proof -- X, Z_i, A_i, sign, the bit tensors -- is indexed batch-major
(flat index b * dim + j, zkfc.cu:86,119-122), so the batch bits are the HIGH variables of every multilinear table.
Rank r holds rows [r B/N, (r+1) B/N) and all challenges are known before any kernel runs, so:

  * a sumcheck over a table of k variables splits into the k - log N LOW rounds, which each rank runs on its own
    shard with the ordinary kernels (zk_relu_prove on the shard with the challenge vectors truncated to the local
    variables), and the log N TOP rounds over the N values the shards folded to.  Round r < k - log N of the whole
    table has coefficients  sum_rank eq(u_top, rank) * c^(rank)  (the eq weight of a pair factors into its local
    and its rank part); the top rounds are a sumcheck over an N-element table -- the same library call on N elements;
  * partial_me / multilinear evaluations over the batch bits are eq(u_top, rank)-weighted sums of the shards' values;
  * the weight side (W.partial_me, the inner-product sumcheck, the opening: all batch-independent) is split by layer:
    layer i belongs to rank i mod N.

Communication per proof: TWO all-gathers of a few KB..100 KB (the shards' local transcripts and folded vectors; then the
owners' FC results) -- no per-round exchange.  Both phases run through the ordinary recorded / zipped prove path.  The assembled transcript is the single-GPU transcript bit for bit
(tests/test_gpu_shard.py), so the verifier does not know the proof was sharded.
"""
import numpy as np

from .proof import call_order, ceil_log2, layer_shapes

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
_RINV = pow(1 << 256, -1, R_MOD)


def _to_int(limbs):                      # Montgomery limbs -> field value
    return int.from_bytes(np.ascontiguousarray(limbs, dtype=np.uint32).tobytes(), "little") * _RINV % R_MOD


def _to_mont(x):
    return np.frombuffer(((x % R_MOD) * (1 << 256) % R_MOD).to_bytes(32, "little"), dtype=np.uint32).copy()


def eq_weight(u_top, rank):
    """eq(u_top, rank) as a Montgomery-form element: prod_t (bit_t(rank) ? u_t : 1 - u_t), LSB-first."""
    w = 1
    for t, ut in enumerate(np.asarray(u_top).reshape(-1, 8)):
        v = _to_int(ut)
        w = w * (v if (rank >> t) & 1 else (1 - v)) % R_MOD
    return _to_mont(w)


def relu_lens(k):
    return (k + 5, k + 5, k + 4, k + 4, k, k, k)


class Shard:
    """One rank's part of one proof: its rows of the batch, the replicated layers (bench.Workload with rows=...)."""

    def __init__(self, wl, rank, world):
        self.wl, self.rank, self.world, self.zk = wl, rank, world, wl.zk
        self.lw = ceil_log2(world)
        self._chal_fc = {c: i for c, (kind, i) in enumerate(call_order(len(wl.shapes))) if kind == "fc"}   # call index -> layer of the zkFC calls
        if 1 << self.lw != world or wl.B_p * world != (1 << ceil_log2(wl.B_p * world)):
            raise ValueError("world size and the local batch must be powers of two")

    # ---- phase 1: everything that reads this rank's activations
    def local_phase(self, chal):
        """All ReLU calls of the proof on this rank's rows, recorded into ONE transcript (so they run zipped across layers
        like an unsharded proof), plus the batch-side contractions of every zkFC call."""
        from .capi import Transcript
        tr = Transcript(self.zk)
        out = self.local_collect(self.local_enqueue(chal, tr), tr)
        tr.close()
        return out

    def local_enqueue(self, chal, tr):
        """The device side of phase 1, without a host wait: the contractions are launched, the ReLU calls recorded on `tr`, and `tr` is SEALED --
        its launches run while the caller enqueues or collects other proofs (prove_sharded_stream)."""
        wl, zk = self.wl, self.zk
        sh = wl.shapes
        kb_l = ceil_log2(wl.B_p)
        out = []
        relu_lens_seen = []
        tr.reset()
        # the batch-side contractions first (immediate launches: an immediate call settles recorded work, which would cut the ReLU
        # calls' recorded chains apart), then every ReLU call recorded on one transcript, then ONE read
        a_dev = {}
        for c, (kind, i, ch) in enumerate(chal):
            if kind == "fc":
                u_bs = ch[0]
                x_in = wl.A[i - 1] if i > 0 else wl.X
                a_dev[c] = zk.fr_mul_scalar(zk.fr_partial_me(x_in, u_bs[:kb_l], sh[i]["in_p"]), eq_weight(u_bs[kb_l:], self.rank))   # X~(., u_bs) restricted to this rank's rows
        for c, (kind, i, ch) in enumerate(chal):
            s = sh[i]
            if kind == "fc":
                out.append(dict(a=a_dev[c]))
            else:
                k_l = kb_l + ceil_log2(s["out_p"])
                lens = relu_lens(k_l)
                wl.relus[i].prove(wl.Z[i], wl.A[i], [c_[:n] for c_, n in zip(ch, lens)], tr)
                relu_lens_seen.append(3 * (k_l + 5) + 1 + 32 + 3 * (k_l + 4) + 1 + 16 + 3 * k_l + 2)
                out.append(dict(fr=None))
        tr.seal()
        return out, relu_lens_seen

    def local_collect(self, state, tr, keep_on_device=False):
        """The host side of phase 1: ONE read of the sealed transcript (waits for that transcript only).  keep_on_device (a single rank: nothing
        crosses a process boundary): the contracted vectors stay where they are."""
        out, relu_lens_seen = state
        fr, _, _ = tr.read_sealed()
        off = 0
        it = iter(relu_lens_seen)
        for part in out:
            if "fr" in part:
                n = next(it)
                part["fr"] = fr[off:off + n].copy()
                off += n
            elif not keep_on_device:
                part["a"] = part["a"].download()
        return out

    # ---- phase 2 (after the first all-gather): the weight side of the layers this rank owns
    def owner_phase(self, chal, gathered):
        """zkFC::prove of the owned layers through the ordinary recorded path, on BATCH-1 stand-ins: X.partial_me(u_bs)
        is the all-reduced vector a (a 1 x in tensor: with no batch challenge partial_me is the identity), and
        Z~(u_out || u_bs) = (a W)~(u_out), so Z's stand-in is the layer applied to a (un-Montgomery'd for the last layer,
        demo.cu:119).  The inner-product sumcheck and the opening never see the batch."""
        from .capi import Transcript
        tr = Transcript(self.zk)
        out = self.owner_collect(self.owner_enqueue(chal, gathered, tr), tr)
        tr.close()
        return out

    def owner_enqueue(self, chal, gathered, tr):
        """The device side of phase 2 without a host wait (sealed, like local_enqueue)."""
        zk, wl = self.zk, self.wl
        L = len(wl.shapes)
        mine = [c for c, (kind, i, ch) in enumerate(chal) if kind == "fc" and i % self.world == self.rank]
        tr.reset()
        if not mine:
            return mine, []
        keep = []
        for c in mine:                       # the stand-ins first (immediate launches) ...
            _, i, _ = chal[c]
            a = None
            for r in range(self.world):
                ar = gathered[r][c]["a"]
                if isinstance(ar, np.ndarray):
                    ar = zk.upload(ar)
                a = ar if a is None else zk.fr_add(a, ar)
            z = wl.fcs[i].forward(a)
            if i == L - 1:
                z = zk.fr_unmont(z)
            keep.append((a, z))
        for c, (a, z) in zip(mine, keep):    # ... then every owned layer recorded on one transcript: zipped like an unsharded proof
            _, i, (u_bs, u_in, u_out) = chal[c]
            wl.fcs[i].prove(a, z, wl.gens[i], u_bs[:0], u_in, u_out, tr)
        tr.seal()
        return mine, keep                    # `keep` holds the stand-ins alive until the sealed phase has been read

    def owner_collect(self, state, tr):
        mine, _keep = state
        if not mine:
            return {}
        sh = self.wl.shapes
        chal_fc = self._chal_fc
        fr, aff, inf = tr.read_sealed()
        out, of, og = {}, 0, 0
        for c in mine:
            s = sh[chal_fc[c]]
            nf, ng = 3 * ceil_log2(s["in_p"]) + 4, 3 * ceil_log2(s["n_gens"]) + 2
            out[c] = dict(fr=fr[of:of + nf].copy(), aff=aff[og:og + ng].copy(), inf=inf[og:og + ng].copy())
            of += nf
            og += ng
        return out

    # ---- assembly (after the second all-gather): identical on every rank
    def assemble(self, chal, gathered, owned):
        """gathered[r][c]: rank r's phase-1 payload of call c; owned[c]: the owner's FC result of call c.
        A few hundred field operations per ReLU call on values that are already on the host (the gathered payloads): done in Python
        integers -- through the device they were ~30 tiny launches with a host round trip each per call (two thirds of a sharded
        proof's wall time at N = 1)."""
        wl, lw, N = self.wl, self.lw, self.world
        sh = wl.shapes
        kb_l = ceil_log2(wl.B_p)
        frs, affs, infs = [], [], []
        for c, (kind, i, ch) in enumerate(chal):
            if kind == "fc":
                frs.append(owned[c]["fr"])
                affs.append(owned[c]["aff"])
                infs.append(owned[c]["inf"])
                continue
            s = sh[i]
            k_l = kb_l + ceil_log2(s["out_p"])
            parts = [gathered[r][c]["fr"] for r in range(N)]
            off = 0
            pieces = []

            def combine(n_vals, u_top):
                """sum_rank eq(u_top, rank) * part[off : off + n_vals]"""
                if N == 1:
                    return parts[0][off:off + n_vals]
                ws = [_to_int(eq_weight(u_top, r)) for r in range(N)]
                return _ints_to_mont([sum(ws[r] * _to_int(parts[r][off + j]) for r in range(N)) % R_MOD for j in range(n_vals)])

            for (u, v, kk, nb) in ((ch[0], ch[1], k_l + 5, 32), (ch[2], ch[3], k_l + 4, 16)):
                pieces.append(combine(3 * kk, u[kk:]))                                   # the low rounds' coefficients
                off += 3 * kk
                finals = [_to_int(parts[r][off]) for r in range(N)]                      # each shard's table folded to one value
                off += 1
                pieces.append(_ints_to_mont(_top_rounds(finals, None, u[kk:], v[kk:])))  # the log N top rounds + a(0)
                pieces.append(combine(nb, ch[4][k_l:]))                                  # partial_me(u_rec, nb)
                off += nb
            pieces.append(combine(3 * k_l, ch[5][k_l:]))                                 # Hadamard: low rounds
            off += 3 * k_l
            fa = [_to_int(parts[r][off]) for r in range(N)]
            fb = [_to_int(parts[r][off + 1]) for r in range(N)]
            pieces.append(_ints_to_mont(_top_rounds(fa, fb, ch[5][k_l:], ch[6][k_l:])))
            frs.append(np.concatenate(pieces))
        return np.concatenate(frs), np.concatenate(affs), np.concatenate(infs)


def _ints_to_mont(vals):
    return np.stack([_to_mont(v) for v in vals]) if len(vals) else np.zeros((0, 8), np.uint32)


def _top_rounds(a, b, u_top, v_top):
    """The log N top rounds of an eq-weighted sumcheck over the N values the shards folded to, as field integers: binary
    (b is None: proof.cu:152-188, sum_g eq(u, g)(a_g^2 - a_g)) or Hadamard (proof.cu:110-137, sum_g eq(u, g) a_g b_g); the weight of
    pair j in round r is eq(u[r+1:], j); both tables fold with v[r].  Returns the 3 log N coefficients followed by a(0) [, b(0)]."""
    us = [_to_int(x) for x in np.asarray(u_top).reshape(-1, 8)]
    vs = [_to_int(x) for x in np.asarray(v_top).reshape(-1, 8)]
    out = []
    a = list(a)
    b = None if b is None else list(b)
    for r in range(len(us)):
        half = len(a) // 2
        w = [1]
        for ut in us[r + 1:]:
            w = [x * (1 - ut) % R_MOD for x in w] + [x * ut % R_MOD for x in w]
        c0 = c1 = c2 = 0
        for j in range(half):
            a0, a1 = a[2 * j], a[2 * j + 1]
            da = a1 - a0
            if b is None:
                c0 += w[j] * (a0 * a0 - a0)
                c1 += w[j] * (2 * a0 * da - da)
                c2 += w[j] * da * da
            else:
                b0, b1 = b[2 * j], b[2 * j + 1]
                db = b1 - b0
                c0 += w[j] * a0 * b0
                c1 += w[j] * (a0 * db + b0 * da)
                c2 += w[j] * da * db
        out += [c0 % R_MOD, c1 % R_MOD, c2 % R_MOD]
        a = [(a[2 * j] + vs[r] * (a[2 * j + 1] - a[2 * j])) % R_MOD for j in range(half)]
        if b is not None:
            b = [(b[2 * j] + vs[r] * (b[2 * j + 1] - b[2 * j])) % R_MOD for j in range(half)]
    out.append(a[0] % R_MOD)
    if b is not None:
        out.append(b[0] % R_MOD)
    return out


def prove_sharded(shards, chal, allgather=None):
    """Run one sharded proof.  `shards`: this process's Shard objects -- all N of them when the ranks are simulated in one
    process (tests), exactly one under torch.distributed, in which case `allgather(obj) -> [obj of rank 0, ..., N-1]`
    is the collective (zkdl_amd.shard.dist_allgather).  Returns (Fr (n, 8), affine G1 (n, 24), infinity flags)."""
    if allgather is None:
        assert len(shards) == shards[0].world
        gathered = [sh.local_phase(chal) for sh in shards]
        owned = {}
        for sh in shards:
            owned.update(sh.owner_phase(chal, gathered))
        return shards[0].assemble(chal, gathered, owned)
    (sh,) = shards
    gathered = allgather(sh.local_phase(chal))
    owned = {}
    for part in allgather(sh.owner_phase(chal, gathered)):
        owned.update(part)
    return sh.assemble(chal, gathered, owned)


def prove_sharded_stream(shard, chals, allgather=None, depth=2):
    """K sharded proofs, software-pipelined: while proof k's first phase is being read, proof k + 1's first phase and proof k - 1's second phase
    are already on the device (every phase is a SEALED transcript: a read waits for that transcript only), so the three host reads per proof
    no longer leave the GPU idle.  `shard`: this process's Shard; `allgather` as in prove_sharded (None: a single rank, payloads stay on the device).
    Yields (Fr, affine G1, infinity flags) per proof, in order -- each the single-GPU transcript of that proof's challenges, bit for bit."""
    from .capi import Transcript
    single = allgather is None
    if single:
        if shard.world != 1:
            raise ValueError("prove_sharded_stream without a collective is the one-rank form")
        allgather = lambda obj: [obj]
    # `depth` proofs between a phase's submission and its read: 2 * depth + 1 proofs in flight (each phase on its own is latency-bound -- 2.5 and 3-4 ms
    # for ~3 ms of chip work -- so the chip fills only when several overlap)
    inflight = 2 * depth + 1
    trsA = [Transcript(shard.zk) for _ in range(inflight)]
    trsB = [Transcript(shard.zk) for _ in range(inflight)]
    chals = list(chals)
    K = len(chals)
    stA, stB, gathered = {}, {}, {}
    # the phases' immediate calls (batch-side contractions, stand-ins) read static tensors, host challenges and what THIS loop has already read back, and
    # write buffers that only later-recorded launches consume: they need main-stream order only, not a join of every worker stream ("api_join" off) --
    # with the join each of them waited for the Fr phases of the proofs in flight, and the stream of proofs ran one proof at a time
    shard.zk.set_flag("api_join", 0)
    try:
        for k in range(K + 2 * depth):
            if k < K:
                stA[k] = shard.local_enqueue(chals[k], trsA[k % inflight])
            j = k - depth
            if 0 <= j < K:
                gathered[j] = allgather(shard.local_collect(stA.pop(j), trsA[j % inflight], keep_on_device=single))
                stB[j] = shard.owner_enqueue(chals[j], gathered[j], trsB[j % inflight])
            j = k - 2 * depth
            if 0 <= j < K:
                owned = {}
                for part in allgather(shard.owner_collect(stB.pop(j), trsB[j % inflight])):
                    owned.update(part)
                yield shard.assemble(chals[j], gathered.pop(j), owned)
    finally:
        shard.zk.set_flag("api_join", 1)
        for t in trsA + trsB:
            t.close()


def dist_allgather(obj):
    """all_gather of a picklable payload over the default process group (RCCL on GPUs: the byte tensors travel through
    device memory; gloo on CPU).  Payloads are a few KB to ~100 KB per rank."""
    import torch.distributed as dist
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
