"""ctypes binding of the CPU oracle (liboracle.so) and, when present, of the reference's own
compiled arithmetic (oracle/_ref/libzkdl_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  Nothing under zkdl_amd/ imports this module.

Arrays are numpy uint32: Fr (n, 8), Fp (n, 12), G1 affine (n, 24), G1 Jacobian (n, 36), in the
reference's limb order (bls12-381.cuh:116-124,218-226,419-430).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_u32 = ctypes.c_uint32


def build(force=False):
    """Compile liboracle.so (and _ref/ when /root/reference is present). Building the checker is
    not using it."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("zkdl_oracle.c", "zkdl_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/bls12-381.cu"):
        ref = os.path.join(_HERE, "_ref", "libzkdl_ref.so")
        if force or not os.path.exists(ref):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _p(a):
    return a.ctypes.data_as(_vp)


def fr(n):
    return np.zeros((n, 8), dtype=np.uint32)


def g1j(n):
    return np.zeros((n, 36), dtype=np.uint32)


def g1a(n):
    return np.zeros((n, 24), dtype=np.uint32)


def _c(a, w=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    if w is not None:
        a = a.reshape(-1, w)
    return a


class Oracle:
    def __init__(self):
        build()
        self.lib = ctypes.CDLL(os.path.join(_HERE, "liboracle.so"))
        L = self.lib
        for name in ("zo_fr_partial_me_size",):
            getattr(L, name).restype = _sz
        L.zo_ceil_log2.restype = _u32
        L.zo_ceil_log2.argtypes = [_u32]
        L.zo_fr_partial_me_size.argtypes = [_sz, _sz, _sz]

    # ---- constants
    def constants(self):
        o = [fr(1), fr(1), fr(1), np.zeros((1, 12), np.uint32), np.zeros((1, 12), np.uint32), np.zeros((1, 12), np.uint32)]
        self.lib.zo_constants(*[_p(x) for x in o])
        return dict(zip(("fr_one", "fr_p", "fr_r2", "fp_one", "fp_p", "fp_r2"), [x[0] for x in o]))

    def generator(self):
        g = g1j(1)
        self.lib.zo_g1_generator(_p(g))
        return g

    def ceil_log2(self, x):
        return int(self.lib.zo_ceil_log2(int(x)))

    # ---- elementwise helpers
    def _bin(self, fn, a, b, w):
        a, b = _c(a, w), _c(b, w)
        out = np.zeros_like(a)
        getattr(self.lib, fn)(_p(a), _p(b), _p(out), _sz(a.shape[0]))
        return out

    def _un(self, fn, a, w):
        a = _c(a, w)
        out = np.zeros_like(a)
        getattr(self.lib, fn)(_p(a), _p(out), _sz(a.shape[0]))
        return out

    def fr_add(self, a, b): return self._bin("zo_fr_vadd", a, b, 8)
    def fr_sub(self, a, b): return self._bin("zo_fr_vsub", a, b, 8)
    def fr_mul(self, a, b): return self._bin("zo_fr_vmul", a, b, 8)
    def fr_neg(self, a): return self._un("zo_fr_vneg", a, 8)
    def fr_mont(self, a): return self._un("zo_fr_vmont", a, 8)
    def fr_unmont(self, a): return self._un("zo_fr_vunmont", a, 8)

    def _bs(self, fn, a, x, w, xw):
        a, x = _c(a, w), _c(x, xw)
        out = np.zeros_like(a)
        getattr(self.lib, fn)(_p(a), _p(x), _p(out), _sz(a.shape[0]))
        return out

    def fr_add_s(self, a, x): return self._bs("zo_fr_vadd_s", a, x, 8, 8)
    def fr_sub_s(self, a, x): return self._bs("zo_fr_vsub_s", a, x, 8, 8)
    def fr_mul_s(self, a, x): return self._bs("zo_fr_vmul_s", a, x, 8, 8)

    def fr_sum(self, a):
        a = _c(a, 8)
        out = fr(1)
        self.lib.zo_fr_sum(_p(a), _sz(a.shape[0]), _p(out))
        return out[0]

    def fr_me(self, t, u):
        t, u = _c(t, 8), _c(u, 8)
        out = fr(1)
        rc = self.lib.zo_fr_me(_p(t), _sz(t.shape[0]), _p(u), _sz(u.shape[0]), _p(out))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out[0]

    def fr_partial_me(self, t, u, window):
        t, u = _c(t, 8), _c(u, 8)
        n = int(self.lib.zo_fr_partial_me_size(t.shape[0], u.shape[0], window))
        out = fr(max(n, 1))
        rc = self.lib.zo_fr_partial_me(_p(t), _sz(t.shape[0]), _p(u), _sz(u.shape[0]), _sz(window), _p(out))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out[:n]

    def fr_split(self, t, window):
        t = _c(t, 8)
        m = (t.shape[0] + 1) // 2
        o0, o1 = fr(m), fr(m)
        rc = self.lib.zo_fr_split(_p(t), _sz(t.shape[0]), _sz(window), _p(o0), _p(o1))
        if rc:
            raise ValueError("Invalid window size.")
        return o0, o1

    # ---- sumchecks
    def ip_sumcheck(self, a, b, u):
        a, b, u = _c(a, 8), _c(b, 8), _c(u, 8)
        if a.shape[0] != b.shape[0]:
            raise ValueError("Incompatible dimensions")
        k = u.shape[0]
        out = fr(3 * k + 2)
        rc = self.lib.zo_inner_product_sumcheck(_p(a), _p(b), _sz(a.shape[0]), _p(u), _sz(k), _p(out))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out

    def hp_sumcheck(self, a, b, u, v):
        a, b, u, v = _c(a, 8), _c(b, 8), _c(u, 8), _c(v, 8)
        if a.shape[0] != b.shape[0] or u.shape[0] != v.shape[0]:
            raise ValueError("Incompatible dimensions")
        k = u.shape[0]
        out = fr(3 * k + 2)
        rc = self.lib.zo_hadamard_product_sumcheck(_p(a), _p(b), _sz(a.shape[0]), _p(u), _p(v), _sz(k), _p(out))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out

    def bin_sumcheck(self, a, u, v):
        a, u, v = _c(a, 8), _c(u, 8), _c(v, 8)
        if u.shape[0] != v.shape[0]:
            raise ValueError("Incompatible dimensions")
        k = u.shape[0]
        out = fr(3 * k + 1)
        rc = self.lib.zo_binary_sumcheck(_p(a), _sz(a.shape[0]), _p(u), _p(v), _sz(k), _p(out))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out

    # ---- G1
    def g1_add(self, a, b): return self._bin("zo_g1_vadd", a, b, 36)
    def g1_sub(self, a, b): return self._bin("zo_g1_vsub", a, b, 36)
    def g1_neg(self, a): return self._un("zo_g1_vneg", a, 36)
    def g1_add_s(self, a, x): return self._bs("zo_g1_vadd_s", a, x, 36, 36)
    def g1_sub_s(self, a, x): return self._bs("zo_g1_vsub_s", a, x, 36, 36)

    def g1_madd(self, a, b):
        a, b = _c(a, 36), _c(b, 24)
        out = np.zeros_like(a)
        self.lib.zo_g1_vmadd(_p(a), _p(b), _p(out), _sz(a.shape[0]))
        return out

    def g1_msub(self, a, b):
        a, b = _c(a, 36), _c(b, 24)
        out = np.zeros_like(a)
        self.lib.zo_g1_vmsub(_p(a), _p(b), _p(out), _sz(a.shape[0]))
        return out

    def g1_madd_s(self, a, x): return self._bs("zo_g1_vmadd_s", a, x, 36, 24)
    def g1_msub_s(self, a, x): return self._bs("zo_g1_vmsub_s", a, x, 36, 24)

    def g1_double(self, a):
        a = _c(a, 36)
        out = np.zeros_like(a)
        for i in range(a.shape[0]):
            self.lib.zo_g1_double(_p(a[i]), _p(out[i]))
        return out

    def g1_from_affine(self, a):
        a = _c(a, 24)
        out = g1j(a.shape[0])
        self.lib.zo_g1_from_affine(_p(a), _p(out), _sz(a.shape[0]))
        return out

    def g1_sum(self, a):
        a = _c(a, 36)
        out = g1j(1)
        self.lib.zo_g1_sum(_p(a), _sz(a.shape[0]), _p(out))
        return out[0]

    def g1_mul(self, g, s):
        g, s = _c(g, 36), _c(s, 8)
        out = g1j(g.shape[0])
        self.lib.zo_g1_vmul(_p(g), _p(s), _p(out), _sz(g.shape[0]))
        return out

    def g1_mul_bcast(self, g, s):
        g, s = _c(g, 36), _c(s, 8)
        if s.shape[0] % g.shape[0]:
            raise ValueError("Incompatible dimensions")
        out = g1j(s.shape[0])
        self.lib.zo_g1_vmul_bcast(_p(g), _sz(g.shape[0]), _p(s), _sz(s.shape[0]), _p(out))
        return out

    def g1_me(self, t, u):
        t, u = _c(t, 36), _c(u, 8)
        out = g1j(1)
        rc = self.lib.zo_g1_me(_p(t), _sz(t.shape[0]), _p(u), _sz(u.shape[0]), _p(out))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out[0]

    def g1_to_affine(self, a):
        a = _c(a, 36)
        out = g1a(a.shape[0])
        inf = np.zeros(a.shape[0], dtype=np.uint8)
        self.lib.zo_g1_to_affine(_p(a), _sz(a.shape[0]), _p(out), _p(inf))
        return out, inf

    # ---- commitment
    def commit(self, gens, t):
        gens, t = _c(gens, 36), _c(t, 8)
        if t.shape[0] % gens.shape[0]:
            raise ValueError("Incompatible dimensions")
        out = g1j(t.shape[0] // gens.shape[0])
        rc = self.lib.zo_commit(_p(gens), _sz(gens.shape[0]), _p(t), _sz(t.shape[0]), _p(out))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out

    def commit_as_launched(self, gens, t):
        gens, t = _c(gens, 36), _c(t, 8)
        m = t.shape[0] // gens.shape[0]
        out = g1j(m)
        written = np.zeros(m, dtype=np.uint8)
        rc = self.lib.zo_commit_as_launched(_p(gens), _sz(gens.shape[0]), _p(t), _sz(t.shape[0]), _p(out), _p(written))
        if rc:
            raise ValueError("Incompatible dimensions")
        return out, written

    def me_open(self, t, gens, u):
        t, gens, u = _c(t, 8), _c(gens, 36), _c(u, 8)
        if t.shape[0] != gens.shape[0]:
            raise ValueError("Incompatible dimensions")
        k = u.shape[0]
        proof = g1j(3 * k + 1)
        ret = fr(1)
        rc = self.lib.zo_me_open(_p(t), _p(gens), _sz(t.shape[0]), _p(u), _sz(k), _p(proof), _p(ret))
        if rc:
            raise ValueError("Incompatible dimensions")
        return ret[0], proof

    def open(self, gens, t, com, u):
        gens, t, com, u = _c(gens, 36), _c(t, 8), _c(com, 36), _c(u, 8)
        k = u.shape[0]
        k_lo = k - self.ceil_log2(com.shape[0])
        com_eval = g1j(1)
        proof = g1j(3 * max(k_lo, 0) + 1)
        ret = fr(1)
        rc = self.lib.zo_open(_p(gens), _sz(gens.shape[0]), _p(t), _sz(t.shape[0]), _p(com), _sz(com.shape[0]),
                              _p(u), _sz(k), _p(com_eval), _p(proof), _p(ret))
        if rc:
            raise ValueError("Incompatible dimensions")
        return ret[0], com_eval[0], proof

    # ---- layers
    def float_to_fr(self, fs, rows_p, cols_p):
        fs = np.ascontiguousarray(fs, dtype=np.float32)
        rows, cols = fs.shape
        out = fr(rows_p * cols_p)
        self.lib.zo_float_to_fr(_p(fs), _p(out), _u32(rows), _u32(rows_p), _u32(cols), _u32(cols_p))
        return out

    def fc_forward(self, X, W, batch, n_in, n_out):
        X, W = _c(X, 8), _c(W, 8)
        Z = fr(batch * n_out)
        self.lib.zo_fc_forward(_p(X), _p(W), _u32(batch), _u32(n_in), _u32(n_out), _p(Z))
        return Z

    def relu_forward(self, X):
        X = _c(X, 8)
        n = X.shape[0]
        Z, sign, mag, rem = fr(n), fr(n), fr(32 * n), fr(16 * n)
        self.lib.zo_relu_forward(_p(X), _sz(n), _p(Z), _p(sign), _p(mag), _p(rem))
        return Z, sign, mag, rem

    def fc_prove(self, X, Z, W, n_in, n_out, gens, com, u_bs, u_in, u_out):
        X, Z, W, gens, com = _c(X, 8), _c(Z, 8), _c(W, 8), _c(gens, 36), _c(com, 36)
        u_bs, u_in, u_out = _c(u_bs, 8), _c(u_in, 8), _c(u_out, 8)
        fr_out = fr(3 * u_in.shape[0] + 8)
        g1_out = g1j(3 * 32 + 8)
        nf, ng = _sz(0), _sz(0)
        rc = self.lib.zo_fc_prove(_p(X), _sz(X.shape[0]), _p(Z), _sz(Z.shape[0]), _p(W), _u32(n_in), _u32(n_out),
                                  _p(gens), _sz(gens.shape[0]), _p(com), _sz(com.shape[0]),
                                  _p(u_bs), _sz(u_bs.shape[0]), _p(u_in), _sz(u_in.shape[0]), _p(u_out), _sz(u_out.shape[0]),
                                  _p(fr_out), ctypes.byref(nf), _p(g1_out), ctypes.byref(ng))
        if rc:
            raise ValueError("Incompatible dimensions (%d)" % rc)
        return fr_out[:nf.value], g1_out[:ng.value]

    def relu_prove(self, X, Z, sign, mag, rem, u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp):
        X, Z, sign, mag, rem = _c(X, 8), _c(Z, 8), _c(sign, 8), _c(mag, 8), _c(rem, 8)
        ch = [_c(x, 8) for x in (u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp)]
        n = X.shape[0]
        k = self.ceil_log2(n)
        fr_out = fr(3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 3 * k + 2 + 8)
        nf = _sz(0)
        rc = self.lib.zo_relu_prove(_p(X), _p(Z), _sz(n), _p(sign), _p(mag), _p(rem), *[_p(c) for c in ch],
                                    _p(fr_out), ctypes.byref(nf))
        if rc:
            raise ValueError("Incompatible dimensions")
        return fr_out[:nf.value]

    def leaky_forward(self, X, shift):
        """LeakyReLU with negative slope 2^-shift (extension; shift 0 = identity): (A, sign, mag_bin, rem_bin, low, nsign)."""
        X = _c(X, 8)
        n = X.shape[0]
        Z, sign, mag, rem, low, nsign = fr(n), fr(n), fr(32 * n), fr(16 * n), fr(n), fr(n)
        self.lib.zo_leaky_forward(_p(X), _sz(n), _u32(shift), _p(Z), _p(sign), _p(mag), _p(rem), _p(low), _p(nsign))
        return Z, sign, mag, rem, low, nsign

    def leaky_prove(self, X, Z, sign, mag, rem, low, nsign, u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp, u_lk, v_lk):
        X, Z, sign, mag, rem, low, nsign = [_c(t, 8) for t in (X, Z, sign, mag, rem, low, nsign)]
        ch = [_c(x, 8) for x in (u_z, v_z, u_r, v_r, u_rec, u_hp, v_hp, u_lk, v_lk)]
        n = X.shape[0]
        k = self.ceil_log2(n)
        fr_out = fr(3 * (k + 5) + 1 + 32 + 3 * (k + 4) + 1 + 16 + 2 * (3 * k + 2) + 8)
        nf = _sz(0)
        rc = self.lib.zo_leaky_prove(_p(X), _p(Z), _sz(n), _p(sign), _p(mag), _p(rem), _p(low), _p(nsign), *[_p(c) for c in ch], _p(fr_out), ctypes.byref(nf))
        if rc:
            raise ValueError("Incompatible dimensions")
        return fr_out[:nf.value]

    def num_threads(self):
        return int(self.lib.zo_num_threads())

    # ---- single-element primitives (same call shape as RefPrims, for limb-for-limb comparison)
    def _call(self, fn, outw, *args):
        n = args[0].shape[0]
        out = np.zeros((n, outw), dtype=np.uint32)
        f = getattr(self.lib, fn)
        for i in range(n):
            f(*[_p(a[i]) for a in args], _p(out[i]))
        return out

    def fr(self, op, *args): return self._call("zo_fr_" + op, 8, *[_c(a, 8) for a in args])
    def fp(self, op, *args): return self._call("zo_fp_" + op, 12, *[_c(a, 12) for a in args])
    def g1_add_mixed(self, a, b): return self._call("zo_g1_add_mixed", 36, _c(a, 36), _c(b, 24))
    def g1_add1(self, a, b): return self._call("zo_g1_add", 36, _c(a, 36), _c(b, 36))
    def g1_mul1(self, a, x): return self._call("zo_g1_mul", 36, _c(a, 36), _c(x, 8))


class RefPrims:
    """The reference's own compiled arithmetic (oracle/_ref). Only available where it was built."""

    PATH = os.path.join(_HERE, "_ref", "libzkdl_ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.PATH)

    def __init__(self):
        self.lib = ctypes.CDLL(self.PATH)

    def _call(self, fn, outw, *args):
        n = args[0].shape[0]
        out = np.zeros((n, outw), dtype=np.uint32)
        f = getattr(self.lib, fn)
        for i in range(n):
            f(*[_p(a[i]) for a in args], _p(out[i]))
        return out

    def fr(self, op, *args): return self._call("ref_fr_" + op, 8, *[_c(a, 8) for a in args])
    def fp(self, op, *args): return self._call("ref_fp_" + op, 12, *[_c(a, 12) for a in args])
    def g1_double(self, a): return self._call("ref_g1_double", 36, _c(a, 36))
    def g1_add(self, a, b): return self._call("ref_g1_add", 36, _c(a, 36), _c(b, 36))
    def g1_add_mixed(self, a, b): return self._call("ref_g1_add_mixed", 36, _c(a, 36), _c(b, 24))

    def constants(self):
        o = [fr(1), fr(1), fr(1), np.zeros((1, 12), np.uint32), np.zeros((1, 12), np.uint32), np.zeros((1, 12), np.uint32)]
        self.lib.ref_constants(*[_p(x) for x in o])
        return dict(zip(("fr_one", "fr_p", "fr_r2", "fp_one", "fp_p", "fp_r2"), [x[0] for x in o]))
