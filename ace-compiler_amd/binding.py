"""ctypes binding of libacehip.so (include/acehip.h).  No arithmetic here: every call lands in the
HIP library; if the library is missing or there is no GPU the calls raise (no CPU fallback)."""
import ctypes as C
import os

import numpy as np

from .build import LIB, build, source_fingerprint

_u32, _u64, _vp, _i32 = C.c_uint32, C.c_uint64, C.c_void_p, C.c_int32


class AceHipError(RuntimeError):
    pass


# (name, restype, argtypes) for every symbol declared in include/acehip.h
SYMBOLS = [
    ("acehip_source_fingerprint", C.c_char_p, []),
    ("acehip_last_error", C.c_char_p, []),
    ("acehip_device_count", C.c_int, []),
    ("acehip_ctx_create", _vp, [_u32, _u32, _u32, _u32, _u32, C.c_int]),
    ("acehip_ctx_destroy", None, [_vp]),
    ("acehip_ctx_create_host", _vp, [_u32, _u32, _u32, _u32, _u32]),
    ("acehip_degree", _u32, [_vp]),
    ("acehip_num_q", _u32, [_vp]),
    ("acehip_num_p", _u32, [_vp]),
    ("acehip_num_q_parts", _u32, [_vp]),
    ("acehip_part_size", _u32, [_vp]),
    ("acehip_num_decomp", _u32, [_vp, _u32]),
    ("acehip_prime", _u64, [_vp, _u32]),
    ("acehip_get_table", C.c_int64, [_vp, C.c_int, _u32, _vp, C.c_size_t]),
    ("acehip_get_modup_tables", C.c_int, [_vp, _u32, _u32, _vp, _vp, _vp, _vp]),
    ("acehip_auto_index", _u32, [_vp, _i32]),
    ("acehip_auto_order", _vp, [_vp, _u32]),
    ("acehip_auto_order_host", C.c_int, [_vp, _u32, _vp]),
    ("acehip_malloc", _vp, [C.c_size_t]),
    ("acehip_free", C.c_int, [_vp]),
    ("acehip_memcpy_h2d", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("acehip_memcpy_d2h", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("acehip_memcpy_d2d", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("acehip_memset", C.c_int, [_vp, C.c_int, C.c_size_t, _vp]),
    ("acehip_malloc_limbs", _vp, [_vp, _vp, C.c_size_t]),
    ("acehip_limb_memory", None, [_vp, _vp]),
    ("acehip_malloc_host", _vp, [C.c_size_t]),
    ("acehip_free_host", C.c_int, [_vp]),
    ("acehip_memcpy_h2d_async", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("acehip_stream_sync", C.c_int, [_vp]),
    ("acehip_event_create", _vp, []),
    ("acehip_event_record", C.c_int, [_vp, _vp]),
    ("acehip_event_elapsed_ms", C.c_int, [_vp, _vp, C.POINTER(C.c_float)]),
    ("acehip_event_sync", C.c_int, [_vp]),
    ("acehip_event_destroy", C.c_int, [_vp]),
    ("acehip_ntt_forward", C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_ntt_inverse", C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_ntt_batch", C.c_int, [_vp, _vp, C.c_size_t, _u32, _u32, _u32, _u32, C.c_int, _vp]),
    ("acehip_modadd", C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_modsub", C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_modmul", C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_modmuladd", C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_rotate", C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_rotate_add2", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    ("acehip_hw_modadd", C.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_hw_modmul", C.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_hw_rotate", C.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_hw_batch", C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    ("acehip_hw_batch_plan", C.c_long, [_vp, _vp, C.c_size_t, _vp, _vp, _vp, C.c_size_t, _u64]),
    ("acehip_hw_batch_discard", C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    ("acehip_hw_batch_plan_discard", C.c_long, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, _vp, C.c_size_t, _u64]),
    ("acehip_stats", C.c_int, [_vp, C.c_int, C.c_int]),
    ("acehip_stat_name", C.c_char_p, [C.c_int]),
    ("acehip_decomp_modup", C.c_int, [_vp, _vp, _vp, _u32, _u32, _vp]),
    ("acehip_mod_down", C.c_int, [_vp, _vp, _vp, _u32, _vp]),
    ("acehip_rescale", C.c_int, [_vp, _vp, _vp, _u32, _vp]),
    ("acehip_mod_down2", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_rescale2", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_mod_raise", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_base_conv", C.c_int, [_vp, _vp, _vp, _u32, C.c_int, _vp, _u32, _vp]),
    ("acehip_key_switch", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_key_switch_bytes", _u64, [_vp, _u32]),
    ("acehip_conv_mfma_tables", C.c_long, [_vp, _u32, C.c_int32, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    ("acehip_values_to_rns", C.c_int, [_vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_encode", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_size_t, _u32, C.c_double, _u32, _u32, _u32, _vp]),
    ("acehip_encode_with_scale", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_size_t, _u32, C.c_double, _u32, _u32, _vp]),
    ("acehip_encode_batch", C.c_int, [_vp, _vp, _vp, _u32, C.c_int, C.c_size_t, _u32, C.c_double, _u32, _u32, _vp]),
    ("acehip_encode_status", C.c_int, [_vp]),
    ("acehip_sample_uniform", C.c_int, [_vp, _vp, _u32, _u32, _u32, _u64, _vp]),
    ("acehip_sample_uniform_keyed", C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp]),
    ("acehip_mul_scalars", C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_add_scalars", C.c_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_modup_digits", C.c_int, [_vp, _vp, _vp, _u32, _vp]),
    ("acehip_modup_digits_to", C.c_int, [_vp, _vp, _vp, _u32, _vp]),
    ("acehip_key_inner_product", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_keymac_mod_down2", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp]),
    ("acehip_keymac_fusable", C.c_int, [_vp, _u32, _u32]),
    ("acehip_debug_set_kmac_fuse", C.c_int, [C.c_int]),
    ("acehip_key_inner_product_add", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    ("acehip_key_inner_products", C.c_int, [_vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _vp, _vp]),
    ("acehip_decomp", C.c_int, [_vp, _vp, _vp, _u32, _u32, _vp]),
    ("acehip_mod_up", C.c_int, [_vp, _vp, _vp, _u32, _u32, _vp]),
    ("acehip_bsgs_inner", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    ("acehip_bsgs_inner_rot", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    ("acehip_shard_create", _vp, [_vp, _u32, _u32]),
    ("acehip_shard_destroy", None, [_vp]),
    ("acehip_shard_num_q", _u32, [_vp, _u32]),
    ("acehip_shard_num_p", _u32, [_vp]),
    ("acehip_shard_pad_q", _u32, [_vp, _u32]),
    ("acehip_shard_pad_p", _u32, [_vp]),
    ("acehip_shard_owned", _u32, [_vp, _u32, _vp, _vp]),
    ("acehip_shard_ks_phase1", C.c_int, [_vp, _vp, _vp, _u32, _vp]),
    ("acehip_shard_ks_phase2", C.c_int, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_shard_ks_phase3", C.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_shard_rescale_send", C.c_int, [_vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_shard_rescale_apply", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("acehip_encode_message", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_size_t, _u32, C.c_double, _vp]),
    ("acehip_shard_encode_limbs", C.c_int, [_vp, _vp, _vp, C.c_double, _u32, _u32, _vp]),
    ("acehip_debug_touches", C.c_size_t, [C.c_int, _vp, _vp, C.c_size_t]),
    # replicas of the caller's arena (image batches, simulated ranks)
    ("acehip_ctx_set_arena", C.c_int, [_vp, _vp]),
    ("acehip_workspace_words", C.c_size_t, [_vp]),
    ("acehip_ctx_select", C.c_int, [_vp, _u32, _u32]),
    ("acehip_upload", C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    ("acehip_download", C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    ("acehip_fill", C.c_int, [_vp, _vp, C.c_int, C.c_size_t, _vp]),
    ("acehip_copy", C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    # limb-sharded execution as a mode of the context
    ("acehip_ctx_shard_sim", C.c_int, [_vp, _u32]),
    ("acehip_rccl_unique_id", C.c_int, [_vp, C.c_size_t]),
    ("acehip_ctx_shard_rccl", C.c_int, [_vp, _u32, _u32, _vp, C.c_size_t]),
    ("acehip_shard_gather", C.c_int, [_vp, _vp, _u32, _u32, _u32, _vp]),
    ("acehip_shard_world", _u32, [_vp]),
    ("acehip_shard_rank", _u32, [_vp]),
    ("acehip_shard_owned_limbs", _u32, [_vp, _u32]),
    ("acehip_shard_traffic", _u64, [_vp, _vp, C.c_int]),
    ("acehip_shard_collectives", _u64, [_vp]),
    ("acehip_shard_schedule", C.c_int, [_vp, _u32, C.c_int, _u32, _vp, _vp, _vp, C.c_size_t]),
]


class ArenaCfg(C.Structure):
    """acehip_arena_cfg of include/acehip.h"""
    _fields_ = [("base", C.c_void_p), ("bytes", C.c_size_t), ("stride_bytes", C.c_size_t), ("n_replicas", C.c_uint32),
                ("workspace", C.c_void_p), ("hw_scratch", C.c_void_p), ("hw_scratch_limbs", C.c_size_t)]

_lib = None


def load_library(path=None):
    """dlopen libacehip.so and declare every prototype; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB
    if path is None:
        build()  # (re)builds when the library is missing or carries another fingerprint than the sources beside it
    lib = C.CDLL(p)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if path is None:  # the in-tree library must have been built from the in-tree sources: never run a stale binary silently
        got, want = lib.acehip_source_fingerprint().decode(), source_fingerprint()
        if got != want:
            raise RuntimeError("libacehip.so was built from other sources (fingerprint %s) than ace-compiler_amd/csrc + include "
                               "(%s): rebuild with `python -m ace_compiler_amd.build --force`" % (got, want))
    if path is None:
        _lib = lib
    return lib


class HwOp(C.Structure):
    """acehip_hw_op of include/acehip.h"""
    _fields_ = [("op", C.c_uint32), ("prime_gi", C.c_uint32), ("res", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p)]


class HwRange(C.Structure):
    """acehip_hw_range of include/acehip.h"""
    _fields_ = [("ptr", C.c_void_p), ("words", C.c_size_t)]


HW_NOSTORE = 0x80000000


HW_ADD, HW_MUL, HW_ROTATE, HW_COPY, HW_ZERO, HW_SUB, HW_MULADD, HW_MULC, HW_ADDC = range(9)


class DeviceBuffer:
    """HBM allocation owned through the C ABI (acehip_malloc/free)."""

    def __init__(self, rt, n_words, dtype=np.uint64):
        self.rt, self.n, self.dtype = rt, int(n_words), np.dtype(dtype)
        self.nbytes = self.n * self.dtype.itemsize
        self.ptr = rt.lib.acehip_malloc(self.nbytes)
        if not self.ptr:
            raise AceHipError(rt.err())

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.size == self.n
        self.rt.check(self.rt.lib.acehip_memcpy_h2d(self.ptr, arr.ctypes.data, self.nbytes, None))
        return self

    def download(self, shape=None):
        out = np.empty(self.n, dtype=self.dtype)
        self.rt.check(self.rt.lib.acehip_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes, None))
        return out.reshape(shape) if shape is not None else out

    def at(self, word_offset):
        return self.ptr + int(word_offset) * self.dtype.itemsize

    def free(self):
        if self.ptr:
            self.rt.lib.acehip_free(self.ptr)
            self.ptr = None


class AceHip:
    """One CKKS context in HBM (acehip_ctx) plus convenience wrappers that move numpy arrays through
    the C ABI.  `host_only=True` builds the tables without a GPU (launches then fail loudly)."""

    def __init__(self, N, L, q0_bits, sf_bits, dnum, device=0, host_only=False):
        self.lib = load_library()
        if host_only:
            self.h = self.lib.acehip_ctx_create_host(N, L, q0_bits, sf_bits, dnum)
        else:
            self.h = self.lib.acehip_ctx_create(N, L, q0_bits, sf_bits, dnum, device)
        if not self.h:
            raise AceHipError(self.err())
        self.N, self.L, self.K = N, self.lib.acehip_num_q(self.h), self.lib.acehip_num_p(self.h)
        self.sf_bits = sf_bits
        self.dnum, self.alpha = self.lib.acehip_num_q_parts(self.h), self.lib.acehip_part_size(self.h)
        self.primes = [self.lib.acehip_prime(self.h, i) for i in range(self.L + self.K)]

    def err(self):
        return (self.lib.acehip_last_error() or b"").decode()

    def check(self, rc):
        if rc != 0:
            raise AceHipError("acehip error %d: %s" % (rc, self.err()))

    def close(self):
        if self.h:
            self.lib.acehip_ctx_destroy(self.h)
            self.h = None

    # ---- tables (host copies) ----
    def table(self, what, gi=0, n=None):
        n = n if n is not None else max(self.N, self.L * self.L, self.L * max(self.K, 1), self.L + self.K)
        out = np.zeros(n, dtype=np.uint64)
        got = self.lib.acehip_get_table(self.h, what, gi, out.ctypes.data, n)
        if got < 0:
            raise AceHipError(self.err())
        return out[:got]

    def modup_tables(self, level, digit):
        hat_inv = np.zeros(64, dtype=np.uint64)
        compl = np.zeros(128, dtype=np.uint32)
        hat_mod = np.zeros(64 * 128, dtype=np.uint64)
        nc = _u32()
        n2 = self.lib.acehip_get_modup_tables(self.h, level, digit, hat_inv.ctypes.data, compl.ctypes.data,
                                              hat_mod.ctypes.data, C.byref(nc))
        if n2 < 0:
            raise AceHipError(self.err())
        return n2, hat_inv[:n2].copy(), compl[: nc.value].copy(), hat_mod[: n2 * nc.value].reshape(n2, nc.value).copy()

    def num_decomp(self, level):
        return self.lib.acehip_num_decomp(self.h, level)

    def auto_index(self, rot_idx):
        return self.lib.acehip_auto_index(self.h, rot_idx)

    def auto_order_host(self, k):
        out = np.zeros(self.N, dtype=np.uint32)
        self.check(self.lib.acehip_auto_order_host(self.h, k, out.ctypes.data))
        return out

    # ---- device helpers ----
    def buf(self, n_words, dtype=np.uint64):
        return DeviceBuffer(self, n_words, dtype)

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, arr.size, arr.dtype).upload(arr)

    def sync(self, stream=None):
        self.check(self.lib.acehip_stream_sync(stream))

    def time_ms(self, fn, reps, stream=None):
        """average device time (ms) of fn() over `reps` back-to-back calls, HIP events on `stream`."""
        e0, e1 = self.lib.acehip_event_create(), self.lib.acehip_event_create()
        self.check(self.lib.acehip_event_record(e0, stream))
        for _ in range(reps):
            fn()
        self.check(self.lib.acehip_event_record(e1, stream))
        ms = C.c_float()
        self.check(self.lib.acehip_event_elapsed_ms(e0, e1, C.byref(ms)))
        self.lib.acehip_event_destroy(e0)
        self.lib.acehip_event_destroy(e1)
        return ms.value / reps

    # ---- numpy-in / numpy-out wrappers (used by tests; each runs the HIP path) ----
    def ntt(self, x, level, pos0=0, inverse=False):
        d = self.to_device(x)
        fn = self.lib.acehip_ntt_inverse if inverse else self.lib.acehip_ntt_forward
        # x holds limbs [pos0, pos0+n): pass a base pointer such that position pos0 is the first limb
        self.check(fn(self.h, d.ptr - pos0 * self.N * 8, level, pos0, x.shape[0], None))
        out = d.download(x.shape)
        d.free()
        return out

    def ew(self, name, a, b, level, pos0=0, acc=None):
        da, db = self.to_device(a), self.to_device(b)
        dr = self.to_device(acc) if acc is not None else self.buf(a.size)
        off = pos0 * self.N * 8
        self.check(getattr(self.lib, "acehip_" + name)(self.h, dr.ptr - off, da.ptr - off, db.ptr - off, level, pos0,
                                                       a.shape[0], None))
        out = dr.download(a.shape)
        for d in (da, db, dr):
            d.free()
        return out

    def rotate(self, a, k, level, pos0=0):
        perm = self.lib.acehip_auto_order(self.h, k)
        if not perm:
            raise AceHipError(self.err())
        da, dr = self.to_device(a), self.buf(a.size)
        off = pos0 * self.N * 8
        self.check(self.lib.acehip_rotate(self.h, dr.ptr - off, da.ptr - off, perm, level, pos0, a.shape[0], None))
        out = dr.download(a.shape)
        da.free()
        dr.free()
        return out

    def decomp_modup(self, a, level, digit):
        da, dr = self.to_device(a), self.buf((level + self.K) * self.N)
        self.check(self.lib.acehip_memset(dr.ptr, 0, dr.nbytes, None))
        self.check(self.lib.acehip_decomp_modup(self.h, dr.ptr, da.ptr, level, digit, None))
        out = dr.download((level + self.K, self.N))
        da.free()
        dr.free()
        return out

    def mod_down(self, ext, level):
        da, dr = self.to_device(ext), self.buf(level * self.N)
        self.check(self.lib.acehip_mod_down(self.h, dr.ptr, da.ptr, level, None))
        out = dr.download((level, self.N))
        da.free()
        dr.free()
        return out

    def rescale(self, a, level):
        da, dr = self.to_device(a), self.buf((level - 1) * self.N)
        self.check(self.lib.acehip_rescale(self.h, dr.ptr, da.ptr, level, None))
        out = dr.download((level - 1, self.N))
        da.free()
        dr.free()
        return out

    def encode(self, values, level, slots=0, sf_degree=1, n_p=0, sf_bits=None):
        """acehip_encode: message (float32 / float64 / complex128 array) -> (q limbs [level,N], p limbs [n_p,N])"""
        v = np.ascontiguousarray(values)
        kind = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.complex128): 2}[v.dtype]
        dv = DeviceBuffer(self, max(v.size, 1), v.dtype)
        if v.size:
            dv.upload(v)
        dq, dp = self.buf(level * self.N), self.buf(max(n_p, 1) * self.N)
        sf = float(2 ** (self.sf_bits if sf_bits is None else sf_bits))
        self.check(self.lib.acehip_encode(self.h, dq.ptr, dp.ptr if n_p else None, dv.ptr, kind, v.size, slots, sf, sf_degree,
                                          level, n_p, None))
        self.check(self.lib.acehip_encode_status(self.h))
        q, p_ = dq.download((level, self.N)), dp.download((max(n_p, 1), self.N))[:n_p]
        for d in (dv, dq, dp):
            d.free()
        return q, p_

    def hw_batch(self, ops):
        """ops: iterable of (op, prime_gi, res_ptr, a_ptr, b_ptr) device addresses -> acehip_hw_batch"""
        arr = (HwOp * len(ops))(*[HwOp(o, g, r, a or None, b or None) for o, g, r, a, b in ops])
        self.check(self.lib.acehip_hw_batch(self.h, arr, len(ops), None))

    def hw_batch_discard(self, ops, dead):
        """acehip_hw_batch_discard: dead = iterable of (device address, words) the caller does not need afterwards"""
        arr = (HwOp * len(ops))(*[HwOp(o, g, r, a or None, b or None) for o, g, r, a, b in ops])
        rg = (HwRange * max(1, len(dead)))(*[HwRange(p, w) for p, w in dead])
        self.check(self.lib.acehip_hw_batch_discard(self.h, arr, len(ops), rg, len(dead), None))

    def key_switch(self, a, key, level):
        da, dk = self.to_device(a), self.to_device(key)
        d0, d1 = self.buf(level * self.N), self.buf(level * self.N)
        self.check(self.lib.acehip_key_switch(self.h, d0.ptr, d1.ptr, da.ptr, dk.ptr, level, None))
        o0, o1 = d0.download((level, self.N)), d1.download((level, self.N))
        for d in (da, dk, d0, d1):
            d.free()
        return o0, o1
