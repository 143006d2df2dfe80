"""ctypes binding of oracle/liboracle.so (the CPU restatement) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")

u64 = C.c_uint64
u32 = C.c_uint32
P64 = C.POINTER(u64)


class OrcPrime(C.Structure):
    _fields_ = [("q", u64), ("br_k", u64), ("br_m", u64), ("prec128_lo", u64), ("prec128_hi", u64),
                ("psi", u64), ("n_inv", u64), ("n_inv_prec", u64), ("rou", P64), ("rou_prec", P64),
                ("rou_inv", P64), ("rou_inv_prec", P64)]


class OrcCtx(C.Structure):
    _fields_ = [("N", u32), ("logN", u32), ("L", u32), ("K", u32), ("dnum", u32), ("alpha", u32),
                ("q0_bits", u32), ("sf_bits", u32), ("prime", C.POINTER(OrcPrime)),
                ("phat_inv_modp", P64), ("phat_inv_modp_prec", P64), ("phat_modq", P64), ("pinv_modq", P64),
                ("ql_inv_modqi", P64), ("ql_inv_modqi_prec", P64), ("qlql", P64), ("qlql_prec", P64)]


def build():
    """(Re)build liboracle.so with gcc if it is missing or older than its sources."""
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ckks_oracle.c", "ckks_encode.c", "ckks_oracle.h", "rou_table.inc")]
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.orc_ctx_create.restype = C.POINTER(OrcCtx)
        L.orc_ctx_create.argtypes = [u32, u32, u32, u32, u32]
        L.orc_ctx_create_from_primes.restype = C.POINTER(OrcCtx)
        L.orc_ctx_create_from_primes.argtypes = [u32, vp, u32, u32]
        L.orc_ctx_free.argtypes = [C.POINTER(OrcCtx)]
        for name in ("orc_mul_mod", "orc_pow_mod"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [u64, u64, u64]
        L.orc_inv_mod_prime.restype = u64
        L.orc_inv_mod_prime.argtypes = [u64, u64]
        L.orc_is_prime.restype = C.c_int
        L.orc_is_prime.argtypes = [u64]
        L.orc_find_generator.restype = u64
        L.orc_find_generator.argtypes = [u64]
        L.orc_root_of_unity.restype = u64
        L.orc_root_of_unity.argtypes = [u64, u64]
        L.orc_find_automorphism_index.restype = u32
        L.orc_find_automorphism_index.argtypes = [C.c_int32, u32]
        L.orc_automorphism_order.argtypes = [vp, u32, u32, C.c_int]
        L.orc_precompute_const_128.argtypes = [u64, P64, P64]
        L.orc_num_decomp.restype = u32
        L.orc_num_decomp.argtypes = [C.POINTER(OrcCtx), u32]
        L.orc_modup_tables.restype = u32
        L.orc_modup_tables.argtypes = [C.POINTER(OrcCtx), u32, u32, vp, vp, vp]
        L.orc_ntt_fwd.argtypes = [vp, C.POINTER(OrcPrime), u32]
        L.orc_ntt_inv.argtypes = [vp, C.POINTER(OrcPrime), u32]
        L.orc_hw_modadd.argtypes = [vp, vp, vp, u64, u32]
        L.orc_hw_modmul.argtypes = [vp, vp, vp, C.POINTER(OrcPrime), u32]
        L.orc_hw_modmul_faithful.argtypes = [vp, vp, vp, C.POINTER(OrcPrime), u32]
        L.orc_hw_rotate.argtypes = [vp, vp, vp, u64, u32]
        L.orc_decomp_modup.argtypes = [C.POINTER(OrcCtx), vp, vp, u32, u32]
        L.orc_mod_down.argtypes = [C.POINTER(OrcCtx), vp, vp, u32]
        L.orc_rescale.argtypes = [C.POINTER(OrcCtx), vp, vp, u32]
        L.orc_key_switch.argtypes = [C.POINTER(OrcCtx), vp, vp, vp, vp, u32]
        L.orc_encode.restype = C.c_int
        L.orc_encode.argtypes = [C.POINTER(OrcCtx), vp, vp, vp, C.c_size_t, u32, u32, u32, u32]
        L.orc_encode_value.restype = C.c_int
        L.orc_encode_value.argtypes = [C.POINTER(OrcCtx), vp, C.c_double, u32, u32]
        L.orc_sum64.restype = u64
        L.orc_sum64.argtypes = [vp, C.c_size_t]
        L.orc_xorw.restype = u64
        L.orc_xorw.argtypes = [vp, C.c_size_t]
        L.orc_splitmix64.restype = u64
        L.orc_splitmix64.argtypes = [u64, u64]
        L.orc_ctx_create_from_primes.restype = C.POINTER(OrcCtx)
        L.orc_ctx_create_from_primes.argtypes = [u32, P64, u32, u32]
        L.orc_switch_modulus.restype = u64
        L.orc_switch_modulus.argtypes = [u64, u64, u64]
        L.orc_mul_mod.restype = u64
        L.orc_mul_mod.argtypes = [u64, u64, u64]
        _lib = L
    return _lib


def encode_message(n, seed):
    """the message oracle/ref_dump.c `encode` feeds the reference: (float)(((i*7+seed)%17)-8)/16"""
    return (((np.arange(n, dtype=np.int64) * 7 + seed) % 17) - 8).astype(np.float32) / np.float32(16.0)


def ptr(a):
    assert a.dtype in (np.uint64, np.int64, np.uint32) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def splitmix64(seed, idx):
    """Vectorised splitmix64(seed, i) identical to orc_splitmix64 / ref_dump.c."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def sum64(a):
    with np.errstate(over="ignore"):
        return int(np.sum(a.reshape(-1).astype(np.uint64), dtype=np.uint64))


def xorw(a):
    a = a.reshape(-1).astype(np.uint64)
    with np.errstate(over="ignore"):
        w = a * (np.uint64(2) * np.arange(a.size, dtype=np.uint64) + np.uint64(1))
    return int(np.bitwise_xor.reduce(w))


class Oracle:
    """One parameter set (ORC_CTX) of the CPU restatement."""

    def __init__(self, N, L, q0_bits, sf_bits, dnum):
        self.lib = lib()
        self.h = self.lib.orc_ctx_create(N, L, q0_bits, sf_bits, dnum)
        c = self.h.contents
        self.N, self.L, self.K, self.dnum, self.alpha = c.N, c.L, c.K, c.dnum, c.alpha
        self.primes = [c.prime[i].q for i in range(self.L + self.K)]

    @classmethod
    def from_primes(cls, N, q_primes, dnum):
        """a context over a GIVEN q chain (the reference's default chains of its unit tests); the p primes are derived"""
        self = cls.__new__(cls)
        self.lib = lib()
        arr = (C.c_uint64 * len(q_primes))(*q_primes)
        self.h = self.lib.orc_ctx_create_from_primes(N, arr, len(q_primes), dnum)
        c = self.h.contents
        self.N, self.L, self.K, self.dnum, self.alpha = c.N, c.L, c.K, c.dnum, c.alpha
        self.primes = [c.prime[i].q for i in range(self.L + self.K)]
        return self

    def close(self):
        if self.h:
            self.lib.orc_ctx_free(self.h)
            self.h = None

    def prime(self, gi):
        return self.h.contents.prime[gi]

    def prime_ptr(self, gi):
        return C.pointer(self.h.contents.prime[gi])

    def table(self, name, n):
        return np.ctypeslib.as_array(getattr(self.h.contents, name), shape=(n,)).copy()

    def twiddles(self, gi, which="rou"):
        return np.ctypeslib.as_array(getattr(self.prime(gi), which), shape=(self.N,)).copy()

    def gidx(self, limb, level):
        """global prime index of limb `limb` of a poly extended at `level` (q then p)."""
        return limb if limb < level else self.L + (limb - level)

    def uniform(self, n_limbs, level, seed):
        """limb l<level mod q_l, limbs >= level mod p_{l-level}: same as ref_dump.c fill_uniform."""
        N = self.N
        out = np.empty((n_limbs, N), dtype=np.uint64)
        for l in range(n_limbs):
            idx = np.arange(l * N, (l + 1) * N, dtype=np.uint64)
            out[l] = splitmix64(seed, idx) % np.uint64(self.primes[self.gidx(l, level)])
        return out

    def ntt_fwd(self, x, gis):
        y = np.ascontiguousarray(x.copy())
        for l, gi in enumerate(gis):
            self.lib.orc_ntt_fwd(ptr(y[l]), self.prime_ptr(gi), self.N)
        return y

    def ntt_inv(self, x, gis):
        y = np.ascontiguousarray(x.copy())
        for l, gi in enumerate(gis):
            self.lib.orc_ntt_inv(ptr(y[l]), self.prime_ptr(gi), self.N)
        return y

    def hw_modadd(self, a, b, gis):
        r = np.empty_like(a)
        for l, gi in enumerate(gis):
            self.lib.orc_hw_modadd(ptr(r[l]), ptr(a[l]), ptr(b[l]), self.primes[gi], self.N)
        return r

    def hw_modmul(self, a, b, gis):
        r = np.empty_like(a)
        for l, gi in enumerate(gis):
            self.lib.orc_hw_modmul(ptr(r[l]), ptr(a[l]), ptr(b[l]), self.prime_ptr(gi), self.N)
        return r

    def automorphism(self, k, is_ntt=True):
        out = np.empty(self.N, dtype=np.int64)
        self.lib.orc_automorphism_order(ptr(out), k, self.N, 1 if is_ntt else 0)
        return out

    def hw_rotate(self, a, perm, gis):
        r = np.empty_like(a)
        for l, gi in enumerate(gis):
            self.lib.orc_hw_rotate(ptr(r[l]), ptr(a[l]), ptr(perm), self.primes[gi], self.N)
        return r

    def decomp_modup(self, a, level, digit):
        out = np.zeros((level + self.K, self.N), dtype=np.uint64)
        self.lib.orc_decomp_modup(self.h, ptr(out), ptr(np.ascontiguousarray(a)), level, digit)
        return out

    def mod_down(self, ext, level):
        out = np.zeros((level, self.N), dtype=np.uint64)
        self.lib.orc_mod_down(self.h, ptr(out), ptr(np.ascontiguousarray(ext)), level)
        return out

    def rescale(self, a, level):
        out = np.zeros((level - 1, self.N), dtype=np.uint64)
        self.lib.orc_rescale(self.h, ptr(out), ptr(np.ascontiguousarray(a)), level)
        return out

    def key_switch(self, a, key, level):
        o0 = np.zeros((level, self.N), dtype=np.uint64)
        o1 = np.zeros((level, self.N), dtype=np.uint64)
        self.lib.orc_key_switch(self.h, ptr(o0), ptr(o1), ptr(np.ascontiguousarray(a)),
                                ptr(np.ascontiguousarray(key)), level)
        return o0, o1

    def encode(self, values, level, slots=0, sf_degree=1, n_p=0):
        """values: complex array (zero padded to slots) -> (q limbs [level,N], p limbs [n_p,N]) NTT domain"""
        v = np.ascontiguousarray(np.asarray(values, dtype=np.complex128))
        q = np.empty((level, self.N), dtype=np.uint64)
        p = np.empty((max(n_p, 1), self.N), dtype=np.uint64)
        rc = lib().orc_encode(self.h, ptr(q), ptr(p), v.ctypes.data_as(C.c_void_p), v.size, slots, sf_degree, level, n_p)
        if rc != 0:
            raise OverflowError("encode overflow")
        return q, p[:n_p]

    def encode_value(self, value, level, sf_degree=1):
        out = np.empty(level, dtype=np.uint64)
        assert lib().orc_encode_value(self.h, ptr(out), float(value), sf_degree, level) == 0
        return out

    def num_decomp(self, level):
        return self.lib.orc_num_decomp(self.h, level)

    def modup_tables(self, level, digit):
        hat_inv = np.zeros(64, dtype=np.uint64)
        compl = np.zeros(128, dtype=np.uint32)
        hat_mod = np.zeros(64 * 128, dtype=np.uint64)
        n2 = self.lib.orc_modup_tables(self.h, level, digit, ptr(hat_inv), ptr(compl), ptr(hat_mod))
        nc = level - n2 + self.K
        return n2, hat_inv[:n2].copy(), compl[:nc].copy(), hat_mod[: n2 * nc].reshape(n2, nc).copy()

    def make_key(self, seed_base):
        """uniform 'key' [dnum][2][L+K][N] as in ref_dump.c (seed_base + d for each of 2*dnum polys)."""
        key = np.empty((self.dnum * 2, self.L + self.K, self.N), dtype=np.uint64)
        for d in range(self.dnum * 2):
            key[d] = self.uniform(self.L + self.K, self.L, seed_base + d)
        return key.reshape(self.dnum, 2, self.L + self.K, self.N)
