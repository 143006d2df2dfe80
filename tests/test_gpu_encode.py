"""Encode path on the GPU box: Encode_plain_from_float / Encode_val_at_level of the drop-in
shim against golden plaintexts produced by the reference itself (tests/golden/ref_encode_*.json, written by
oracle/ref_dump.c `encode`: Encode_at_level_with_sf ckks_encoder.c:395, Encode_val_at_level :464).
Bit-exact: the FP64 canonical embedding follows the reference's butterfly order with no FMA contraction, the
integer part runs in HIP kernels."""
import ctypes as C
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_encode_*.json")))


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    import ace_compiler_amd  # noqa: F401

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    so = str(tmp_path_factory.mktemp("stub") / "libctxstub.so")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", os.path.join(ROOT, "tests", "c", "ctx_stub.c"), "-I", inc, "-I",
                           os.path.join(inc, "rt_ant"), "-L", bmod.LIBDIR, "-Wl,--no-as-needed", "-lFHErt_ant", "-Wl,-rpath," + bmod.LIBDIR, "-o", so])
    lib = C.CDLL(so, mode=C.RTLD_GLOBAL)
    lib.Stub_set_params.argtypes = [C.c_uint32] + [C.c_size_t] * 5
    lib.Stub_sizeof_plaintext.restype = C.c_size_t
    lib.Stub_plain_data.restype = C.c_void_p
    lib.Stub_plain_data.argtypes = [C.c_void_p]
    lib.Stub_plain_level.restype = C.c_size_t
    lib.Stub_plain_level.argtypes = [C.c_void_p]
    lib.Encode_plain_from_float.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    lib.Encode_plain_from_double.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
    lib.Free_plain.argtypes = [C.c_void_p]
    lib.acehip_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def _download(lib, pt, N):
    lib.Acehip_rt_sync()  # the shim batches per-limb work lazily; raw reads of Coeffs() memory sync first
    level = lib.Stub_plain_level(pt)
    out = np.empty((level, N), dtype=np.uint64)
    assert lib.acehip_memcpy_d2h(out.ctypes.data, lib.Stub_plain_data(pt), out.nbytes, None) == 0
    return out


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[11:-5] for p in FILES])
def test_encode_matches_reference(stub, path):
    g = json.load(open(path))
    N, level = g["N"], g["level"]
    stub.Stub_set_params(N, g["L"] - 1, g["q0_bits"], g["sf_bits"], g["dnum_req"], 192)
    stub.Prepare_context()
    try:
        for case in g["cases"]:
            n = case["len"]
            msg = O.encode_message(n, g["seed"])
            pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
            stub.Encode_plain_from_float(pt, msg.ctypes.data, n, case["sf_degree"], level)
            got = _download(stub, pt, N)
            gold = case["poly"]
            assert got.size == gold["n"]
            if "data" in gold:
                assert got.reshape(-1).tolist() == gold["data"], (n, case["sf_degree"])
            assert O.sum64(got) == gold["sum64"] and O.xorw(got) == gold["xorw"], (n, case["sf_degree"])
            stub.Free_plain(pt)
        for k in g["consts"]:
            val = np.array([k["value"]], dtype=np.float64)
            pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
            stub.Encode_plain_from_double(pt, val.ctypes.data, 1, k["sf_degree"], level)
            got = _download(stub, pt, N)
            assert got[:, 0].tolist() == k["limb0"], k
            assert np.all(got == got[:, :1])  # constant polynomial in the NTT domain
            stub.Free_plain(pt)
    finally:
        stub.Finalize_context()


# ---- acehip_encode through the C ABI against the oracle (oracle/ckks_encode.c, itself pinned to the reference
# fixtures by tests/test_oracle_golden.py::test_encode_matches_reference) on inputs the fixtures do not cover:
# complex values, sparse slot counts, extended (q+p) plaintexts, ragged and empty messages, wide dynamic range.
ENC_CASES = [
    # N, L, q0, sf, dnum, level, slots, len, kind, sf_degree, n_p
    (16, 3, 60, 50, 2, 3, 0, 8, "f32", 1, 0),
    (16, 3, 60, 50, 2, 2, 2, 2, "c128", 1, 0),
    (16, 3, 60, 50, 2, 3, 1, 1, "f64", 2, 2),
    (64, 7, 60, 51, 3, 4, 0, 0, "f32", 1, 0),          # empty message: the zero plaintext
    (64, 7, 60, 51, 3, 7, 8, 5, "c128", 3, 3),
    (1024, 7, 60, 51, 3, 6, 0, 511, "f64", 1, 0),       # ragged
    (1024, 7, 60, 51, 3, 6, 256, 256, "c128", 1, 3),    # exactly the low-pass block size
    (1024, 7, 60, 51, 3, 6, 512, 300, "c128", 2, 0),    # one strided stage
    (4096, 6, 60, 50, 3, 6, 0, 2048, "f32", 1, 0),
    (65536, 4, 60, 56, 2, 4, 0, 32768, "f32", 1, 0),    # ResNet weight shape: full slots, float message
    (65536, 4, 60, 56, 2, 3, 4096, 4096, "c128", 1, 1), # bootstrap precompute shape: sparse slots, extended
    (131072, 3, 60, 50, 1, 2, 0, 65536, "f64", 1, 0),   # largest ring degree the compiler emits
]


def _message(kind, n, seed):
    rng = np.random.default_rng(seed)
    mag = np.ldexp(rng.standard_normal(n), rng.integers(-20, 4, size=n))   # wide dynamic range
    if kind == "f32":
        return mag.astype(np.float32)
    if kind == "f64":
        return mag.astype(np.float64)
    return (mag + 1j * rng.standard_normal(n)).astype(np.complex128)


@pytest.mark.parametrize("case", ENC_CASES, ids=["N%d_lv%d_s%d_n%d_%s_d%d_p%d" % (c[0], c[5], c[6], c[7], c[8], c[9], c[10]) for c in ENC_CASES])
def test_encode_abi_matches_oracle(case):
    import ace_compiler_amd as A

    N, L, q0, sf, dnum, level, slots, n, kind, deg, n_p = case
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum, device=0)
    try:
        msg = _message(kind, n, 1000 + N + n)
        q, p = rt.encode(msg, level, slots=slots, sf_degree=deg, n_p=n_p)
        eq, ep = o.encode(msg.astype(np.complex128), level, slots=slots, sf_degree=deg, n_p=n_p)
        assert np.array_equal(q, eq)
        assert np.array_equal(p, ep)
    finally:
        rt.close()
        o.close()


def test_encode_overflow_is_reported():
    """|x * Delta| > 9.2e18 is the reference's "encode overflow" assert (ckks_encoder.c:255-258): the device path
    records it and acehip_encode_status fails; the oracle reports the same."""
    import ace_compiler_amd as A

    o = O.Oracle(64, 3, 60, 50, 1)
    rt = A.AceHip(64, 3, 60, 50, 1, device=0)
    try:
        msg = np.full(32, 1.0e5, dtype=np.float64)
        with pytest.raises(OverflowError):
            o.encode(msg.astype(np.complex128), 3)
        with pytest.raises(A.AceHipError, match="encode overflow"):
            rt.encode(msg, 3)
        q, _ = rt.encode(np.ones(32), 3)  # the flag is cleared once reported
        assert np.array_equal(q, o.encode(np.ones(32, dtype=np.complex128), 3)[0])
    finally:
        rt.close()
        o.close()


def test_two_host_threads_own_independent_contexts(stub):
    """All runtime state is per host thread (context, pool, queue, HIP stream): two threads that prepare a context each and
    encode concurrently must both reproduce the reference plaintexts (the bench runs 4 such image streams per GPU)."""
    import threading

    path = [p for p in FILES if "n1024" in p][0]
    g = json.load(open(path))
    N, level = g["N"], g["level"]
    stub.Stub_set_params(N, g["L"] - 1, g["q0_bits"], g["sf_bits"], g["dnum_req"], 192)
    errors = []
    gate = threading.Barrier(2)

    def run(tid):
        try:
            stub.Prepare_context()
            gate.wait()
            for _ in range(20):
                for case in g["cases"]:
                    msg = O.encode_message(case["len"], g["seed"])
                    pt = C.create_string_buffer(stub.Stub_sizeof_plaintext())
                    stub.Encode_plain_from_float(pt, msg.ctypes.data, case["len"], case["sf_degree"], level)
                    got = _download(stub, pt, N)
                    if O.sum64(got) != case["poly"]["sum64"] or O.xorw(got) != case["poly"]["xorw"]:
                        errors.append((tid, case["len"], case["sf_degree"]))
                    stub.Free_plain(pt)
            gate.wait()
            stub.Finalize_context()
        except BaseException as e:  # noqa: BLE001
            errors.append((tid, repr(e)))
            gate.abort()

    ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
