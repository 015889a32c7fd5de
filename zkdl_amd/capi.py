"""ctypes binding of libzkdl_amd.so (include/zkdl_amd.h) -- the only way Python reaches the HIP
kernels.  There is no CPU fallback: if the shared library is missing, or no MI355X is visible,
construction raises.

Device arrays are thin handles (`DevArray`) over zk_malloc'd memory; numpy arrays cross with
explicit upload()/download().  Element layouts (uint32 limbs): Fr (n, 8), G1 affine (n, 24),
G1 Jacobian (n, 36) -- byte-compatible with the reference's Fr_t / G1Affine_t / G1Jacobian_t.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libzkdl_amd.so")

_vp = ctypes.c_void_p
_u32 = ctypes.c_uint32
_sz = ctypes.c_size_t

ZK_OK, ZK_ERR_SHAPE, ZK_ERR_ALLOC, ZK_ERR_DEVICE, ZK_ERR_ARG, ZK_ERR_RANGE = range(6)
MSM_LITERAL, MSM_PIPPENGER = 0, 1

# every symbol include/zkdl_amd.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = """zk_ctx_create zk_ctx_destroy zk_ctx_sync zk_last_error zk_version zk_ctx_stream zk_malloc zk_free
zk_memcpy_h2d zk_memcpy_d2h zk_memcpy_d2d zk_memset_zero zk_timer_start zk_timer_stop_ms zk_prof_enable zk_prof_reset
zk_prof_read zk_rng_seed zk_random_vec zk_fr_random zk_fr_random_int zk_ceil_log2 zk_fr_add zk_fr_sub zk_fr_mul zk_fr_neg
zk_fr_add_scalar zk_fr_sub_scalar zk_fr_mul_scalar zk_fr_mont zk_fr_unmont zk_fr_sum zk_fr_me zk_fr_partial_me_size
zk_fr_partial_me zk_fr_split zk_sumcheck_ip zk_sumcheck_hadamard zk_sumcheck_binary zk_g1_fill zk_g1a_fill
zk_g1_from_affine zk_g1_neg zk_g1a_neg zk_g1_add zk_g1_sub zk_g1_madd zk_g1_msub zk_g1_add_bcast zk_g1_sub_bcast
zk_g1_madd_bcast zk_g1_msub_bcast zk_g1_sum zk_g1_mul zk_g1_mul_bcast zk_g1_me zk_g1_to_affine zk_set_msm_mode
zk_set_msm_window zk_set_flag zk_gens_create zk_gens_destroy zk_gens_size zk_gens_points zk_commit zk_me_open zk_open zk_msm_batch zk_float_to_fr zk_fc_create zk_fc_from_float zk_fc_destroy zk_fc_in
zk_fc_out zk_fc_com_size zk_fc_weights zk_fc_com zk_fc_load_input zk_fc_forward zk_fc_prove zk_relu_create
zk_relu_destroy zk_relu_forward zk_relu_sign zk_relu_mag_bin zk_relu_rem_bin zk_relu_prove zk_transcript_create
zk_transcript_destroy zk_transcript_reset zk_transcript_flush zk_transcript_counts zk_transcript_read zk_transcript_read_affine
zk_transcript_digest zk_transcript_digest_dev zk_transcript_serialize zk_transcript_seal zk_transcript_sizes zk_transcript_read_sealed zk_transcript_seal_pair zk_transcript_stream
zk_verify_fc zk_verify_relu zk_verify_open zk_fs_derive zk_sha256 zk_fc_prove_fs zk_relu_prove_fs zk_leaky_prove_fs zk_fs_round_challenges zk_relu_set_activation zk_leaky_prove zk_verify_leaky zk_act_open zk_act_open_fs zk_act_commit zk_verify_act_open zk_dist_unique_id zk_dist_init zk_dist_world_size zk_dist_sum_rows zk_dist_allreduce_u64
zk_dist_lanes_to_fr zk_dist_destroy zk_gens_prepare_full zk_field_selftest""".split()


class ZkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class ShapeError(ZkError, ValueError):
    """The reference's std::runtime_error("Incompatible dimensions ...")."""


_CTYPE = {"int": ctypes.c_int, "uint32_t": _u32, "uint64_t": ctypes.c_uint64, "size_t": _sz, "float": ctypes.c_float}


def _header_prototypes():
    """(name, restype, argtypes) for every function include/zkdl_amd.h declares: pointers of any
    kind -> c_void_p, so a Python int can never be truncated to a 32-bit C int on its way in."""
    import re
    hdr = os.path.join(os.path.dirname(_HERE), "include", "zkdl_amd.h")
    text = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    protos = []
    for ret, name, args in re.findall(r"([A-Za-z_][\w \*]*?)\b(zk_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret = ret.strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else _vp
        else:
            restype = _CTYPE.get(ret.replace("const", "").strip(), ctypes.c_int)
        argtypes = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_char_p if "char" in a and "**" not in a else _vp)
                else:
                    base = a.replace("const", "").split()[0]
                    argtypes.append(_CTYPE[base])
        protos.append((name, restype, argtypes))
    return protos


_LIB = None


def load_library():
    """The shared library with its prototypes set (loaded once per process)."""
    global _LIB
    if _LIB is None:
        _LIB = _load_library()
    return _LIB


def _load_library():
    # ZKDL_AMD_LIB=<path>: another build of the SAME library (tools/ab_phases.sh points it at lib/variants/lib_exp.so, the -DZKDL_EXPERIMENTS build,
    # instead of swapping files); never the oracle, never a CPU fallback
    path = os.environ.get("ZKDL_AMD_LIB") or LIB_PATH          # (empty = the shipping library)
    if path != LIB_PATH:
        if not os.path.exists(path):
            raise RuntimeError("ZKDL_AMD_LIB=%s does not exist" % path)
        sys.stderr.write("zkdl_amd: loading %s instead of the shipping library\n" % path)
        lib = ctypes.CDLL(path)
        for name, restype, argtypes in _header_prototypes():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        return lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libzkdl_amd.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "there is no CPU fallback for the proving path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in _header_prototypes():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def _hp(a):
    return a.ctypes.data_as(_vp)


def _fr_host(a, k=None):
    a = np.ascontiguousarray(a, dtype=np.uint32).reshape(-1, 8)
    if k is not None and a.shape[0] != k:
        raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
    if a.shape[0] == 0:
        a = np.zeros((1, 8), dtype=np.uint32)
    return a


class DevArray:
    """Owning handle of a device buffer of `n` elements of `width` uint32 limbs."""

    def __init__(self, zk, n, width, ptr=None, own=True):
        self.zk, self.n, self.width, self.own = zk, int(n), int(width), own
        if ptr is None:
            p = _vp()
            zk._chk(zk.lib.zk_malloc(zk.ctx, _sz(max(self.nbytes, 16)), ctypes.byref(p)))
            ptr = p.value
        self.ptr = ptr

    @property
    def nbytes(self):
        return self.n * self.width * 4

    def download(self):
        out = np.zeros((self.n, self.width), dtype=np.uint32)
        if self.n:
            self.zk._chk(self.zk.lib.zk_memcpy_d2h(self.zk.ctx, _hp(out), _vp(self.ptr), _sz(self.nbytes)))
        return out

    def view(self, off, n):
        return DevArray(self.zk, n, self.width, ptr=self.ptr + off * self.width * 4, own=False)

    def free(self):
        if self.own and self.ptr:
            self.zk.lib.zk_free(self.zk.ctx, _vp(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Zk:
    """One context = one GPU + one HIP stream."""

    def __init__(self, device=0):
        self.lib = load_library()
        ctx = _vp()
        rc = self.lib.zk_ctx_create(int(device), ctypes.byref(ctx))
        if rc != ZK_OK:
            raise RuntimeError("zk_ctx_create(device=%d) failed with status %d: no usable MI355X/HIP device "
                               "(the proving path has no CPU fallback)" % (device, rc))
        self.ctx = ctx
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.zk_ctx_destroy(self.ctx)
            self.ctx = None

    def _chk(self, rc):
        if rc == ZK_OK:
            return
        msg = self.lib.zk_last_error(self.ctx).decode()
        if rc == ZK_ERR_SHAPE:
            raise ShapeError(rc, msg)
        raise ZkError(rc, "zkdl_amd status %d: %s" % (rc, msg))

    # ---- memory
    def upload(self, a, width=None):
        a = np.ascontiguousarray(a)
        if a.dtype != np.uint32:
            raise TypeError("expected uint32 limbs")
        width = width or a.shape[-1]
        a = a.reshape(-1, width)
        d = DevArray(self, a.shape[0], width)
        if a.size:
            self._chk(self.lib.zk_memcpy_h2d(self.ctx, _vp(d.ptr), _hp(a), _sz(a.nbytes)))
        return d

    def upload_f32(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        d = DevArray(self, a.size, 1)
        self._chk(self.lib.zk_memcpy_h2d(self.ctx, _vp(d.ptr), _hp(a), _sz(a.nbytes)))
        return d

    def empty(self, n, width):
        return DevArray(self, n, width)

    def sync(self):
        self._chk(self.lib.zk_ctx_sync(self.ctx))

    def seed(self, s):
        self._chk(self.lib.zk_rng_seed(self.ctx, ctypes.c_uint64(s)))

    def random_vec(self, n):
        out = np.zeros((max(n, 1), 8), dtype=np.uint32)
        self._chk(self.lib.zk_random_vec(self.ctx, _u32(n), _hp(out)))
        return out[:n]

    def fr_random(self, n):
        d = self.empty(n, 8)
        self._chk(self.lib.zk_fr_random(self.ctx, _vp(d.ptr), _u32(n)))
        return d

    def fr_random_int(self, n, bits):
        d = self.empty(n, 8)
        self._chk(self.lib.zk_fr_random_int(self.ctx, _vp(d.ptr), _u32(n), _u32(bits)))
        return d

    # ---- timing / profiling
    def timer_start(self):
        self._chk(self.lib.zk_timer_start(self.ctx))

    def timer_stop_ms(self):
        ms = ctypes.c_float()
        self._chk(self.lib.zk_timer_stop_ms(self.ctx, ctypes.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        self._chk(self.lib.zk_prof_enable(self.ctx, int(on)))

    def prof_reset(self):
        self._chk(self.lib.zk_prof_reset(self.ctx))

    def prof_read(self):
        cap = 64
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        launches = (ctypes.c_uint64 * cap)()
        nbytes = (ctypes.c_uint64 * cap)()
        n = ctypes.c_int()
        self._chk(self.lib.zk_prof_read(self.ctx, names, ms, launches, nbytes, cap, ctypes.byref(n)))
        return {names[i].decode(): {"ms": ms[i], "launches": int(launches[i]), "bytes": int(nbytes[i])} for i in range(n.value)}

    # ---- Fr tensor
    def _fr3(self, fn, a, b):
        if a.n != b.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        out = self.empty(a.n, 8)
        self._chk(getattr(self.lib, fn)(self.ctx, _vp(a.ptr), _vp(b.ptr), _vp(out.ptr), _u32(a.n)))
        return out

    def fr_add(self, a, b): return self._fr3("zk_fr_add", a, b)
    def fr_sub(self, a, b): return self._fr3("zk_fr_sub", a, b)
    def fr_mul(self, a, b): return self._fr3("zk_fr_mul", a, b)

    def _fr2(self, fn, a):
        out = self.empty(a.n, 8)
        self._chk(getattr(self.lib, fn)(self.ctx, _vp(a.ptr), _vp(out.ptr), _u32(a.n)))
        return out

    def field_selftest(self, which, op, a, b=None):
        """zk_field_selftest: the device field functions element-wise (which 0 Fr / 1 Fp / 2 Fr via 29-bit limbs / 3 Fp via 28-bit limbs)."""
        out = self.empty(a.n, a.width)
        self._chk(self.lib.zk_field_selftest(self.ctx, ctypes.c_int(which), ctypes.c_int(op), _vp(a.ptr), _vp(b.ptr if b is not None else 0), _vp(out.ptr), _u32(a.n)))
        return out

    def fr_neg(self, a): return self._fr2("zk_fr_neg", a)
    def fr_mont(self, a): return self._fr2("zk_fr_mont", a)
    def fr_unmont(self, a): return self._fr2("zk_fr_unmont", a)

    def _frs(self, fn, a, x):
        x = _fr_host(x, 1)
        out = self.empty(a.n, 8)
        self._chk(getattr(self.lib, fn)(self.ctx, _vp(a.ptr), _hp(x), _vp(out.ptr), _u32(a.n)))
        return out

    def fr_add_scalar(self, a, x): return self._frs("zk_fr_add_scalar", a, x)
    def fr_sub_scalar(self, a, x): return self._frs("zk_fr_sub_scalar", a, x)
    def fr_mul_scalar(self, a, x): return self._frs("zk_fr_mul_scalar", a, x)

    def fr_sum(self, a):
        out = np.zeros(8, dtype=np.uint32)
        self._chk(self.lib.zk_fr_sum(self.ctx, _vp(a.ptr), _u32(a.n), _hp(out)))
        return out

    def fr_me(self, t, u):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        out = np.zeros(8, dtype=np.uint32)
        self._chk(self.lib.zk_fr_me(self.ctx, _vp(t.ptr), _u32(t.n), _hp(_fr_host(u)), _u32(u.shape[0]), _hp(out)))
        return out

    def fr_partial_me(self, t, u, window):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        n = int(self.lib.zk_fr_partial_me_size(_u32(t.n), _u32(u.shape[0]), _u32(window)))
        out = self.empty(n, 8)
        self._chk(self.lib.zk_fr_partial_me(self.ctx, _vp(t.ptr), _u32(t.n), _hp(_fr_host(u)), _u32(u.shape[0]), _u32(window), _vp(out.ptr)))
        return out

    def fr_split(self, t, window):
        m = (t.n + 1) // 2
        o0, o1 = self.empty(m, 8), self.empty(m, 8)
        self._chk(self.lib.zk_fr_split(self.ctx, _vp(t.ptr), _u32(t.n), _u32(window), _vp(o0.ptr), _vp(o1.ptr)))
        return o0, o1

    # ---- sumchecks
    def sumcheck_ip(self, a, b, u):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        if a.n != b.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        k = u.shape[0]
        out = np.zeros((3 * k + 2, 8), dtype=np.uint32)
        self._chk(self.lib.zk_sumcheck_ip(self.ctx, _vp(a.ptr), _vp(b.ptr), _u32(a.n), _hp(_fr_host(u)), _u32(k), _hp(out)))
        return out

    def sumcheck_hadamard(self, a, b, u, v):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        v = np.ascontiguousarray(v, dtype=np.uint32).reshape(-1, 8)
        if a.n != b.n or u.shape[0] != v.shape[0]:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        k = u.shape[0]
        out = np.zeros((3 * k + 2, 8), dtype=np.uint32)
        self._chk(self.lib.zk_sumcheck_hadamard(self.ctx, _vp(a.ptr), _vp(b.ptr), _u32(a.n), _hp(_fr_host(u)), _hp(_fr_host(v)), _u32(k), _hp(out)))
        return out

    def sumcheck_binary(self, a, u, v):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        v = np.ascontiguousarray(v, dtype=np.uint32).reshape(-1, 8)
        if u.shape[0] != v.shape[0]:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        k = u.shape[0]
        out = np.zeros((3 * k + 1, 8), dtype=np.uint32)
        self._chk(self.lib.zk_sumcheck_binary(self.ctx, _vp(a.ptr), _u32(a.n), _hp(_fr_host(u)), _hp(_fr_host(v)), _u32(k), _hp(out)))
        return out

    # ---- G1 tensor
    def g1_fill(self, n, g):
        g = np.ascontiguousarray(g, dtype=np.uint32).reshape(36)
        out = self.empty(n, 36)
        self._chk(self.lib.zk_g1_fill(self.ctx, _vp(out.ptr), _u32(n), _hp(g)))
        return out

    def g1_from_affine(self, a):
        out = self.empty(a.n, 36)
        self._chk(self.lib.zk_g1_from_affine(self.ctx, _vp(a.ptr), _vp(out.ptr), _u32(a.n)))
        return out

    def g1_neg(self, a):
        out = self.empty(a.n, 36)
        self._chk(self.lib.zk_g1_neg(self.ctx, _vp(a.ptr), _vp(out.ptr), _u32(a.n)))
        return out

    def _g3(self, fn, a, b):
        if a.n != b.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        out = self.empty(a.n, 36)
        self._chk(getattr(self.lib, fn)(self.ctx, _vp(a.ptr), _vp(b.ptr), _vp(out.ptr), _u32(a.n)))
        return out

    def g1_add(self, a, b): return self._g3("zk_g1_add", a, b)
    def g1_sub(self, a, b): return self._g3("zk_g1_sub", a, b)
    def g1_madd(self, a, b): return self._g3("zk_g1_madd", a, b)
    def g1_msub(self, a, b): return self._g3("zk_g1_msub", a, b)

    def _gb(self, fn, a, x, w):
        x = np.ascontiguousarray(x, dtype=np.uint32).reshape(w)
        out = self.empty(a.n, 36)
        self._chk(getattr(self.lib, fn)(self.ctx, _vp(a.ptr), _hp(x), _vp(out.ptr), _u32(a.n)))
        return out

    def g1_add_bcast(self, a, x): return self._gb("zk_g1_add_bcast", a, x, 36)
    def g1_sub_bcast(self, a, x): return self._gb("zk_g1_sub_bcast", a, x, 36)
    def g1_madd_bcast(self, a, x): return self._gb("zk_g1_madd_bcast", a, x, 24)
    def g1_msub_bcast(self, a, x): return self._gb("zk_g1_msub_bcast", a, x, 24)

    def g1_sum(self, a):
        out = np.zeros(36, dtype=np.uint32)
        self._chk(self.lib.zk_g1_sum(self.ctx, _vp(a.ptr), _u32(a.n), _hp(out)))
        return out

    def g1_mul(self, g, s):
        if g.n != s.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions 01")
        out = self.empty(g.n, 36)
        self._chk(self.lib.zk_g1_mul(self.ctx, _vp(g.ptr), _vp(s.ptr), _vp(out.ptr), _u32(g.n)))
        return out

    def g1_mul_bcast(self, g, s):
        out = self.empty(s.n, 36)
        self._chk(self.lib.zk_g1_mul_bcast(self.ctx, _vp(g.ptr), _u32(g.n), _vp(s.ptr), _u32(s.n), _vp(out.ptr)))
        return out

    def g1_me(self, t, u):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        out = np.zeros(36, dtype=np.uint32)
        self._chk(self.lib.zk_g1_me(self.ctx, _vp(t.ptr), _u32(t.n), _hp(_fr_host(u)), _u32(u.shape[0]), _hp(out)))
        return out

    def g1_to_affine(self, a):
        out = self.empty(a.n, 24)
        inf = DevArray(self, (a.n + 3) // 4 + 1, 1)
        self._chk(self.lib.zk_g1_to_affine(self.ctx, _vp(a.ptr), _u32(a.n), _vp(out.ptr), _vp(inf.ptr)))
        flags = inf.download().view(np.uint8).reshape(-1)[:a.n].copy()
        return out.download(), flags

    # ---- commitment
    def set_msm_mode(self, mode):
        self._chk(self.lib.zk_set_msm_mode(self.ctx, int(mode)))

    def set_flag(self, name, value):
        self._chk(self.lib.zk_set_flag(self.ctx, name.encode(), int(value)))

    def set_msm_window(self, bits):
        self._chk(self.lib.zk_set_msm_window(self.ctx, _u32(bits)))

    def gens(self, points):
        """Commitment generators handle (commitment.cuh:8-23) from a DevArray of Jacobian points."""
        return Gens(self, points)

    def commit(self, gens, t):
        if gens.n == 0 or t.n % gens.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        out = self.empty(t.n // gens.n, 36)
        self._chk(self.lib.zk_commit(self.ctx, gens.h, _vp(t.ptr), _u32(t.n), _vp(out.ptr)))
        return out

    def me_open(self, t, gens, u):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        if t.n != gens.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        k = u.shape[0]
        proof = np.zeros((3 * k + 1, 36), dtype=np.uint32)
        ret = np.zeros(8, dtype=np.uint32)
        self._chk(self.lib.zk_me_open(self.ctx, gens.h, _vp(t.ptr), _hp(_fr_host(u)), _u32(k), _hp(proof), _hp(ret)))
        return ret, proof

    def open(self, gens, t, com, u):
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        k = u.shape[0]
        k_lo = max(k - int(self.lib.zk_ceil_log2(_u32(com.n))), 0)
        com_eval = np.zeros(36, dtype=np.uint32)
        proof = np.zeros((3 * k_lo + 1, 36), dtype=np.uint32)
        ret = np.zeros(8, dtype=np.uint32)
        self._chk(self.lib.zk_open(self.ctx, gens.h, _vp(t.ptr), _u32(t.n), _vp(com.ptr), _u32(com.n),
                                   _hp(_fr_host(u)), _u32(k), _hp(com_eval), _hp(proof), _hp(ret)))
        return ret, com_eval, proof

    def msm_batch(self, gens, scalars):
        """q = scalars.n / gens.n independent MSMs over the handle's generators (raw canonical scalars)."""
        if gens.n == 0 or scalars.n % gens.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        q = scalars.n // gens.n
        out = self.empty(q, 36)
        self._chk(self.lib.zk_msm_batch(self.ctx, gens.h, _vp(scalars.ptr), _u32(q), _vp(out.ptr)))
        return out

    # ---- verifier (no witness: public commitment / generators, challenges, proof values)
    def verify_fc(self, gens, com, batch, n_in, n_out, ch, fr, aff, inf, z_is_raw=False, expect_z=None, expect_x=None):
        """zk_verify_fc: returns (ok, message, claim_z, claim_x)."""
        fr = np.ascontiguousarray(fr, dtype=np.uint32).reshape(-1, 8)
        aff = np.ascontiguousarray(aff, dtype=np.uint32).reshape(-1, 24)
        inf = np.ascontiguousarray(inf, dtype=np.uint8).reshape(-1)
        ok = ctypes.c_int(0)
        cz, cx = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
        ez = _hp(_fr_host(expect_z, 1)) if expect_z is not None else None
        ex = _hp(_fr_host(expect_x, 1)) if expect_x is not None else None
        self._chk(self.lib.zk_verify_fc(self.ctx, gens.h, _vp(com.ptr), _u32(com.n), _u32(batch), _u32(n_in), _u32(n_out), int(bool(z_is_raw)),
                                        _hp(_fr_host(ch[0])), _hp(_fr_host(ch[1])), _hp(_fr_host(ch[2])), _hp(fr), _sz(fr.shape[0]), _hp(aff), _hp(inf), _sz(aff.shape[0]),
                                        ez, ex, _hp(cz), _hp(cx), ctypes.byref(ok)))
        return bool(ok.value), ("" if ok.value else self.lib.zk_last_error(self.ctx).decode()), cz, cx

    def verify_open(self, gens, com, u, value, aff, inf):
        """zk_verify_open: an opening of the commitment `com` (DevArray of points) at u with opened value `value`."""
        u = np.ascontiguousarray(u, dtype=np.uint32).reshape(-1, 8)
        aff = np.ascontiguousarray(aff, dtype=np.uint32).reshape(-1, 24)
        inf = np.ascontiguousarray(inf, dtype=np.uint8).reshape(-1)
        ok = ctypes.c_int(0)
        self._chk(self.lib.zk_verify_open(self.ctx, gens.h, _vp(com.ptr), _u32(com.n), _hp(_fr_host(u)), _u32(u.shape[0]), _hp(_fr_host(value, 1)), _hp(aff), _hp(inf),
                                          _sz(aff.shape[0]), ctypes.byref(ok)))
        return bool(ok.value), ("" if ok.value else self.lib.zk_last_error(self.ctx).decode())

    def verify_relu(self, n, ch, fr):
        """zk_verify_relu: ch in zk_relu_prove's order (7 vectors; u_rec is not used by any check).  Returns
        (ok, message, claims[5]) -- the claims are the evaluations the reference leaves unbound."""
        fr = np.ascontiguousarray(fr, dtype=np.uint32).reshape(-1, 8)
        ok = ctypes.c_int(0)
        claims = np.zeros((5, 8), np.uint32)
        self._chk(self.lib.zk_verify_relu(self.ctx, _u32(n), _hp(_fr_host(ch[0])), _hp(_fr_host(ch[1])), _hp(_fr_host(ch[2])), _hp(_fr_host(ch[3])), _hp(_fr_host(ch[5])),
                                          _hp(_fr_host(ch[6])), _hp(fr), _sz(fr.shape[0]), _hp(claims), ctypes.byref(ok)))
        return bool(ok.value), ("" if ok.value else self.lib.zk_last_error(self.ctx).decode()), claims

    def verify_leaky(self, n, ch, fr):
        """zk_verify_leaky: ch = zk_leaky_prove's nine vectors.  Returns (ok, message, claims[8])."""
        fr = np.ascontiguousarray(fr, dtype=np.uint32).reshape(-1, 8)
        ok = ctypes.c_int(0)
        claims = np.zeros((8, 8), np.uint32)
        self._chk(self.lib.zk_verify_leaky(self.ctx, _u32(n), *[_hp(_fr_host(ch[i])) for i in (0, 1, 2, 3, 5, 6, 7, 8)], _hp(fr), _sz(fr.shape[0]), _hp(claims),
                                           ctypes.byref(ok)))
        return bool(ok.value), ("" if ok.value else self.lib.zk_last_error(self.ctx).decode()), claims

    def act_open(self, gens, z, a, pts, alpha, fold, tr):
        """zk_act_open: the three activation-side claims of a hidden layer reduced to one opening of [Z ; A], recorded into tr."""
        if z.n != a.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        k = self.ceil_log2(z.n)
        pts = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint32).reshape(-1, 8) for p in pts]), dtype=np.uint32)
        fold = np.ascontiguousarray(fold, dtype=np.uint32).reshape(-1, 8)
        if pts.shape[0] != 3 * k or fold.shape[0] != k + 1:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        self._chk(self.lib.zk_act_open(self.ctx, gens.h, _vp(z.ptr), _vp(a.ptr), _u32(z.n), _hp(_fr_host(pts)), _hp(_fr_host(alpha, 1)), _hp(fold), tr.h))

    def act_commit(self, gens, t, tr, max_bits=0):
        """zk_act_commit: the row-wise commitment of tensor t over gens, deferred into tr (t.n / gens.n points appended to its G1 list).
        max_bits: a bound on the centred values' bit length (0 = full width); checked on the device (ZK_ERR_RANGE from the next sync)."""
        self._chk(self.lib.zk_act_commit(self.ctx, gens.h, _vp(t.ptr), _u32(t.n), _u32(max_bits), tr.h))

    def act_open_fs(self, gens, z, a, pts, seed32, layer, tr):
        """zk_act_open_fs: the same with alpha and the reduction's round challenges hashed from seed32 and the reduction's own round messages."""
        if z.n != a.n:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        k = self.ceil_log2(z.n)
        pts = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint32).reshape(-1, 8) for p in pts]), dtype=np.uint32)
        seed32 = bytes(seed32)
        if pts.shape[0] != 3 * k or len(seed32) != 32:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        self._chk(self.lib.zk_act_open_fs(self.ctx, gens.h, _vp(z.ptr), _vp(a.ptr), _u32(z.n), _hp(_fr_host(pts)), seed32, _u32(layer), tr.h))

    def verify_act_open(self, gens, com2, k, pts, alpha, fold, ys, fr, aff, inf):
        """zk_verify_act_open: com2 = DevArray of the 2m points [com(Z) ; com(A)]."""
        pts = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.uint32).reshape(-1, 8) for p in pts]), dtype=np.uint32)
        fold = np.ascontiguousarray(fold, dtype=np.uint32).reshape(-1, 8)
        ys = np.ascontiguousarray(np.asarray(ys, dtype=np.uint32).reshape(3, 8))
        fr = np.ascontiguousarray(fr, dtype=np.uint32).reshape(-1, 8)
        aff = np.ascontiguousarray(aff, dtype=np.uint32).reshape(-1, 24)
        inf = np.ascontiguousarray(inf, dtype=np.uint8).reshape(-1)
        ok = ctypes.c_int(0)
        self._chk(self.lib.zk_verify_act_open(self.ctx, gens.h, _vp(com2.ptr), _u32(com2.n), _u32(k), _hp(_fr_host(pts)), _hp(_fr_host(alpha, 1)), _hp(fold), _hp(ys),
                                              _hp(fr), _sz(fr.shape[0]), _hp(aff), _hp(inf), _sz(aff.shape[0]), ctypes.byref(ok)))
        return bool(ok.value), ("" if ok.value else self.lib.zk_last_error(self.ctx).decode())

    # ---- layers
    def float_to_fr(self, fs_d, rows, rows_p, cols, cols_p):
        out = self.empty(rows_p * cols_p, 8)
        self._chk(self.lib.zk_float_to_fr(self.ctx, _vp(fs_d.ptr), _vp(out.ptr), _u32(rows), _u32(rows_p), _u32(cols), _u32(cols_p)))
        return out

    def ceil_log2(self, x):
        return int(self.lib.zk_ceil_log2(_u32(x)))

    def fc_load_input(self, batch, in_dim, x_f32_dev):
        """zkFC::load_float_gpu_input (zkfc.cu:106-115): quantise + zero-pad to powers of two."""
        b_p, in_p = 1 << self.ceil_log2(batch), 1 << self.ceil_log2(in_dim)
        out = self.empty(b_p * in_p, 8)
        self._chk(self.lib.zk_fc_load_input(self.ctx, _u32(batch), _u32(in_dim), _vp(x_f32_dev.ptr), _vp(out.ptr)))
        return out


def sha256(data):
    """SHA-256 through the library (host code; the same function zk_fs_derive hashes with)."""
    lib = load_library()
    out = (ctypes.c_ubyte * 32)()
    buf = bytes(data)
    rc = lib.zk_sha256(buf, _sz(len(buf)), out)
    if rc != ZK_OK:
        raise ZkError(rc, "zk_sha256 failed")
    return bytes(out)


def fs_derive(seed32, layer, kind, vec, n, lib=None):
    """n challenges of one prove call's challenge vector, derived from the statement digest (zk_fs_derive)."""
    lib = lib or load_library()
    seed32 = bytes(seed32)
    if len(seed32) != 32:
        raise ValueError("seed must be 32 bytes")
    out = np.zeros((max(n, 1), 8), dtype=np.uint32)
    rc = lib.zk_fs_derive(seed32, _u32(layer), _u32(kind), _u32(vec), _u32(n), _hp(out))
    if rc != ZK_OK:
        raise ZkError(rc, "zk_fs_derive failed")
    return out[:n]


def fs_round_challenges(seed32, layer, kind, which, coeffs, lib=None):
    """The k round challenges of one sumcheck, re-derived from its coefficient triples (zk_fs_round_challenges; host code)."""
    lib = lib or load_library()
    seed32 = bytes(seed32)
    if len(seed32) != 32:
        raise ValueError("seed must be 32 bytes")
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint32).reshape(-1, 8)
    if coeffs.shape[0] % 3:
        raise ValueError("coefficient triples expected")
    k = coeffs.shape[0] // 3
    out = np.zeros((max(k, 1), 8), dtype=np.uint32)
    src = coeffs if k else np.zeros((3, 8), dtype=np.uint32)
    rc = lib.zk_fs_round_challenges(seed32, _u32(layer), _u32(kind), _u32(which), _hp(src), _u32(k), _hp(out))
    if rc != ZK_OK:
        raise ZkError(rc, "zk_fs_round_challenges failed")
    return out[:k]


class Gens:
    """Commitment object (commitment.cuh:8-23): generator tensor + lazily built MSM window tables."""

    def __init__(self, zk, points):
        self.zk = zk
        h = _vp()
        zk._chk(zk.lib.zk_gens_create(zk.ctx, _vp(points.ptr), _u32(points.n), ctypes.byref(h)))
        self.h = h
        self.n = points.n

    def points(self):
        return DevArray(self.zk, self.n, 36, ptr=self.zk.lib.zk_gens_points(self.h), own=False)

    def prepare_full(self, window_bits=12):
        """Precompute every signed digit of every window (zk_gens_prepare_full): full-width MSMs over these generators then run without buckets."""
        self.zk._chk(self.zk.lib.zk_gens_prepare_full(self.zk.ctx, self.h, _u32(window_bits)))

    def close(self):
        if self.h:
            self.zk.lib.zk_gens_destroy(self.h)
            self.h = None


class Transcript:
    def __init__(self, zk):
        self.zk = zk
        h = _vp()
        zk._chk(zk.lib.zk_transcript_create(zk.ctx, ctypes.byref(h)))
        self.h = h

    def reset(self):
        self.zk._chk(self.zk.lib.zk_transcript_reset(self.h))

    def flush(self):
        self.zk._chk(self.zk.lib.zk_transcript_flush(self.h))

    def counts(self):
        a, b = _sz(), _sz()
        self.zk._chk(self.zk.lib.zk_transcript_counts(self.h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def read(self):
        nf, ng = self.counts()
        fr = np.zeros((max(nf, 1), 8), dtype=np.uint32)
        g1 = np.zeros((max(ng, 1), 36), dtype=np.uint32)
        self.zk._chk(self.zk.lib.zk_transcript_read(self.h, _hp(fr), _sz(fr.shape[0]), _hp(g1), _sz(g1.shape[0])))
        return fr[:nf], g1[:ng]

    def read_affine(self):
        _, ng = self.counts()
        aff = np.zeros((max(ng, 1), 24), dtype=np.uint32)
        inf = np.zeros(max(ng, 1), dtype=np.uint8)
        self.zk._chk(self.zk.lib.zk_transcript_read_affine(self.h, _hp(aff), _hp(inf), _sz(aff.shape[0])))
        return aff[:ng], inf[:ng]

    def digest(self):
        out = np.zeros(8, dtype=np.uint32)
        self.zk._chk(self.zk.lib.zk_transcript_digest(self.h, _hp(out)))
        return out

    def digest_dev(self, out8_ptr):
        """Digest as 8 uint64 lanes written to device memory at `out8_ptr` (no host sync)."""
        self.zk._chk(self.zk.lib.zk_transcript_digest_dev(self.h, _vp(out8_ptr)))

    def serialize(self):
        """The transcript as bytes in the library's on-disk format (include/zkdl_amd.h: zk_transcript_serialize)."""
        n = ctypes.c_size_t(0)
        self.zk.lib.zk_transcript_serialize(self.h, None, ctypes.c_size_t(0), ctypes.byref(n))
        buf = (ctypes.c_ubyte * n.value)()
        self.zk._chk(self.zk.lib.zk_transcript_serialize(self.h, buf, ctypes.c_size_t(n.value), ctypes.byref(n)))
        return bytes(buf)

    def seal(self, out8_ptr=None):
        """Finish the proof on the G1 stream without making the main stream wait (zk_transcript_seal)."""
        self.zk._chk(self.zk.lib.zk_transcript_seal(self.h, _vp(out8_ptr) if out8_ptr else None))

    def read_sealed(self):
        """(Fr, affine G1, infinity flags) of a SEALED transcript, waiting for that transcript only (zk_transcript_read_sealed)."""
        a, b = _sz(), _sz()
        self.zk._chk(self.zk.lib.zk_transcript_sizes(self.h, ctypes.byref(a), ctypes.byref(b)))
        nf, ng = a.value, b.value
        fr = np.zeros((max(nf, 1), 8), dtype=np.uint32)
        aff = np.zeros((max(ng, 1), 24), dtype=np.uint32)
        inf = np.zeros(max(ng, 1), dtype=np.uint8)
        self.zk._chk(self.zk.lib.zk_transcript_read_sealed(self.h, _hp(fr), _sz(fr.shape[0]), _hp(aff), _hp(inf), _sz(aff.shape[0])))
        return fr[:nf], aff[:ng], inf[:ng]

    def serialize_sealed(self):
        """zk_transcript_serialize's bytes for a SEALED transcript, without a context-wide wait (read_sealed + the on-disk framing)."""
        fr, aff, inf = self.read_sealed()
        return b"ZKDLTR01" + int(fr.shape[0]).to_bytes(8, "little") + int(aff.shape[0]).to_bytes(8, "little") + fr.tobytes() + aff.tobytes() + inf.tobytes()

    def seal_pair(self, other, out8_ptr=None, other_out8_ptr=None):
        """This proof and `other` submitted together: one zipped Fr phase, two G1 phases (zk_transcript_seal_pair)."""
        self.zk._chk(self.zk.lib.zk_transcript_seal_pair(self.h, _vp(out8_ptr) if out8_ptr else None, other.h, _vp(other_out8_ptr) if other_out8_ptr else None))

    def stream(self):
        """hipStream_t (as int) the last seal() was enqueued on."""
        return self.zk.lib.zk_transcript_stream(self.h)

    def close(self):
        if self.h:
            self.zk.lib.zk_transcript_destroy(self.h)
            self.h = None


class FC:
    """zkFC (zkfc.cuh:16-33)."""

    def __init__(self, zk, handle):
        self.zk, self.h = zk, handle

    @classmethod
    def create(cls, zk, n_in, n_out, weights, gens):
        h = _vp()
        zk._chk(zk.lib.zk_fc_create(zk.ctx, _u32(n_in), _u32(n_out), _vp(weights.ptr), gens.h, ctypes.byref(h)))
        return cls(zk, h)

    @classmethod
    def from_float(cls, zk, n_in, n_out, w_f32_dev, gens):
        h = _vp()
        zk._chk(zk.lib.zk_fc_from_float(zk.ctx, _u32(n_in), _u32(n_out), _vp(w_f32_dev.ptr), gens.h, ctypes.byref(h)))
        return cls(zk, h)

    @property
    def n_in(self): return int(self.zk.lib.zk_fc_in(self.h))
    @property
    def n_out(self): return int(self.zk.lib.zk_fc_out(self.h))
    @property
    def com_size(self): return int(self.zk.lib.zk_fc_com_size(self.h))

    def weights(self):
        return DevArray(self.zk, self.n_in * self.n_out, 8, ptr=self.zk.lib.zk_fc_weights(self.h), own=False)

    def com(self):
        return DevArray(self.zk, self.com_size, 36, ptr=self.zk.lib.zk_fc_com(self.h), own=False)

    def forward(self, x):
        if x.n % self.n_in:
            raise ShapeError(ZK_ERR_SHAPE, "Incompatible dimensions")
        z = self.zk.empty(x.n // self.n_in * self.n_out, 8)
        self.zk._chk(self.zk.lib.zk_fc_forward(self.zk.ctx, self.h, _vp(x.ptr), _u32(x.n), _vp(z.ptr)))
        return z

    def prove(self, x, z, gens, u_bs, u_in, u_out, tr):
        zk = self.zk
        zk._chk(zk.lib.zk_fc_prove(zk.ctx, self.h, _vp(x.ptr), _u32(x.n), _vp(z.ptr), _u32(z.n), gens.h,
                                   _hp(_fr_host(u_bs)), _hp(_fr_host(u_in)), _hp(_fr_host(u_out)), tr.h))

    def prove_fs(self, x, z, gens, seed32, layer, tr):
        """zkFC::prove with round-by-round Fiat-Shamir challenges (zk_fc_prove_fs)."""
        zk = self.zk
        zk._chk(zk.lib.zk_fc_prove_fs(zk.ctx, self.h, _vp(x.ptr), _u32(x.n), _vp(z.ptr), _u32(z.n), gens.h, bytes(seed32), _u32(layer), tr.h))

    def close(self):
        if self.h:
            self.zk.lib.zk_fc_destroy(self.h)
            self.h = None


class ReLU:
    """zkReLU (zkrelu.cuh:10-20)."""

    def __init__(self, zk):
        self.zk = zk
        h = _vp()
        zk._chk(zk.lib.zk_relu_create(zk.ctx, ctypes.byref(h)))
        self.h = h
        self.n = 0

    def forward(self, x):
        z = self.zk.empty(x.n, 8)
        self.zk._chk(self.zk.lib.zk_relu_forward(self.zk.ctx, self.h, _vp(x.ptr), _u32(x.n), _vp(z.ptr)))
        self.n = x.n
        return z

    def aux(self):
        L = self.zk.lib
        return (DevArray(self.zk, self.n, 8, ptr=L.zk_relu_sign(self.h), own=False),
                DevArray(self.zk, self.n * 32, 8, ptr=L.zk_relu_mag_bin(self.h), own=False),
                DevArray(self.zk, self.n * 16, 8, ptr=L.zk_relu_rem_bin(self.h), own=False))

    def prove(self, x, z, ch, tr):
        zk = self.zk
        hs = [_hp(_fr_host(c)) for c in ch]
        zk._chk(zk.lib.zk_relu_prove(zk.ctx, self.h, _vp(x.ptr), _vp(z.ptr), _u32(x.n), *hs, tr.h))

    def set_activation(self, kind, shift=0):
        """0 = ReLU (default); 1 = LeakyReLU with negative slope 2^-shift (shift 0: identity).  Takes effect at the next forward()."""
        self.zk._chk(self.zk.lib.zk_relu_set_activation(self.h, _u32(kind), _u32(shift)))

    def prove_leaky(self, x, z, ch, tr):
        """ch: the reference's seven vectors followed by u_lk, v_lk (zk_leaky_prove)."""
        zk = self.zk
        hs = [_hp(_fr_host(c)) for c in ch]
        zk._chk(zk.lib.zk_leaky_prove(zk.ctx, self.h, _vp(x.ptr), _vp(z.ptr), _u32(x.n), *hs, tr.h))

    def prove_fs(self, x, z, seed32, layer, tr):
        """zkReLU::prove with round-by-round Fiat-Shamir challenges (zk_relu_prove_fs)."""
        zk = self.zk
        zk._chk(zk.lib.zk_relu_prove_fs(zk.ctx, self.h, _vp(x.ptr), _vp(z.ptr), _u32(x.n), bytes(seed32), _u32(layer), tr.h))

    def prove_leaky_fs(self, x, z, seed32, layer, tr):
        """zk_leaky_prove with round-by-round Fiat-Shamir challenges (zk_leaky_prove_fs)."""
        zk = self.zk
        zk._chk(zk.lib.zk_leaky_prove_fs(zk.ctx, self.h, _vp(x.ptr), _vp(z.ptr), _u32(x.n), bytes(seed32), _u32(layer), tr.h))

    def close(self):
        if self.h:
            self.zk.lib.zk_relu_destroy(self.h)
            self.h = None
