"""CPU tests (no GPU) of the clear tensor arithmetic behind the validation helpers the code generator emits (<op>_ref of the reference's
include/ckks/cipher_valid.h, src/ckks/cipher_valid.c; product: csrc/rt/rt_valid.cpp): exported by libFHErt_ant.so, called through
ctypes, compared with numpy restatements of the same definitions."""
import ctypes as C

import numpy as np
import pytest

import ace_compiler_amd as A


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    import os
    import subprocess
    import sys

    from conftest import ROOT

    bmod = sys.modules["ace_compiler_amd.build"]
    A.load_library()
    bmod.build_rt()
    # libFHErt_ant.so leaves the callbacks of a generated program undefined (Get_context_params ...): tests/c/ctx_stub.c supplies them
    stub = str(tmp_path_factory.mktemp("stub") / "libctxstub.so")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", os.path.join(ROOT, "tests", "c", "ctx_stub.c"), "-I", inc, "-I",
                           os.path.join(inc, "rt_ant"), "-L", bmod.LIBDIR, "-Wl,--no-as-needed", "-lFHErt_ant", "-Wl,-rpath," + bmod.LIBDIR,
                           "-o", stub])
    so = C.CDLL(stub, mode=C.RTLD_GLOBAL)
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    so.Conv_ref.restype = so.Gemm_ref.restype = so.Average_pool_ref.restype = so.Global_average_pool_ref.restype = dp
    so.Relu_ref.restype = so.Add_ref.restype = so.Max_pool_ref.restype = dp
    so.Conv_ref.argtypes = [dp] + [C.c_int] * 4 + [fp] + [C.c_int] * 4 + [fp] + [C.c_int] * 7
    so.Gemm_ref.argtypes = [dp, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, C.c_int]
    so.Average_pool_ref.argtypes = so.Max_pool_ref.argtypes = [dp] + [C.c_int] * 12
    so.Global_average_pool_ref.argtypes = [dp] + [C.c_int] * 4
    so.Relu_ref.argtypes = [dp, C.c_uint64]
    so.Add_ref.argtypes = [dp, dp, C.c_uint64]
    return so


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(C.POINTER(C.c_float))


def _take(ptr, n):
    out = np.ctypeslib.as_array(ptr, shape=(n,)).copy()
    C.CDLL(None).free(ptr)
    return out


def test_every_validation_helper_is_exported(lib):
    for name in ("Validate Add_plain_msg Add_msg Add_ref Mul_plain_msg Mul_msg Rotate_msg Relu_msg Relu_rtv Relu_ref Bootstrap_msg Conv_rtv Conv_ref "
                 "Gemm_rtv Gemm_ref Average_pool_rtv Average_pool_ref Max_pool_rtv Max_pool_ref Global_average_pool_rtv Global_average_pool_ref "
                 "Upscale_ciph Downscale_ciph Real_relu Get_msg_with_imag Print_cipher_info Print_cipher_range Print_cipher_poly "
                 "Print_cipher_msg_with_imag Print_poly_lite Encode_plain_from_float_with_scale Get_dcmplx_msg_from_plain Bootstrap_precom").split():
        assert hasattr(lib, name), name


@pytest.mark.parametrize("pad", [0, 1])
def test_conv_ref(lib, pad):
    rng = np.random.default_rng(3)
    n, c, h, w, kn, kh, kw = 1, 3, 6, 6, 4, 3, 3
    x, wt, b = rng.standard_normal((n, c, h, w)), rng.standard_normal((kn, c, kh, kw)).astype(np.float32), rng.standard_normal(kn).astype(np.float32)
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    oh, ow = h + 2 * pad - kh + 1, w + 2 * pad - kw + 1
    want = np.zeros((n, kn, oh, ow))
    for j in range(kn):
        for y in range(oh):
            for xx in range(ow):
                want[0, j, y, xx] = (xp[0, :, y:y + kh, xx:xx + kw] * wt[j].astype(np.float64)).sum() + b[j]
    got = _take(lib.Conv_ref(_d(x), n, c, h, w, _f(wt), kn, c, kh, kw, _f(b), kn, 1, 1, 0, 0, pad, pad), want.size)
    assert np.allclose(got, want.ravel(), rtol=1e-12, atol=1e-12)


def test_gemm_relu_add_pools(lib):
    rng = np.random.default_rng(4)
    x, wt, b = rng.standard_normal(16), rng.standard_normal((10, 16)).astype(np.float32), rng.standard_normal(10).astype(np.float32)
    assert np.allclose(_take(lib.Gemm_ref(_d(x), 1, 16, _f(wt), 10, 16, _f(b), 10), 10), wt.astype(np.float64) @ x + b, rtol=1e-12)
    v = rng.standard_normal(33)
    assert np.array_equal(_take(lib.Relu_ref(_d(v), 33), 33), np.maximum(v, 0))
    assert np.array_equal(_take(lib.Add_ref(_d(v), _d(2 * v), 33), 33), v + 2 * v)
    t = rng.standard_normal((1, 2, 8, 8))
    want = t.reshape(1, 2, 4, 2, 4, 2).mean(axis=(3, 5))
    assert np.allclose(_take(lib.Average_pool_ref(_d(t), 1, 2, 8, 8, 2, 2, 2, 2, 0, 0, 0, 0), 32), want.ravel(), rtol=1e-12)
    assert np.allclose(_take(lib.Max_pool_ref(_d(t), 1, 2, 8, 8, 2, 2, 2, 2, 0, 0, 0, 0), 32), want.ravel(), rtol=1e-12)  # (validated as average)
    assert np.allclose(_take(lib.Global_average_pool_ref(_d(t), 1, 2, 8, 8), 2), t.mean(axis=(2, 3)).ravel(), rtol=1e-12)


def _chacha_block_py(key, counter, nonce):
    """RFC 8439 section 2.3 on Python integers (the checker of the runtime's block function)"""
    M = 0xFFFFFFFF
    rotl = lambda x, n: ((x << n) | (x >> (32 - n))) & M  # noqa: E731
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key) + [counter] + list(nonce)
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M; x[b] = rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M for a, b in zip(x, s)]


def test_chacha20_block_of_the_random_streams(lib):
    """the block function behind key generation / encryption randomness (csrc/rt/rt_rng.hpp) against the known answer of RFC 8439 2.3.2
    and against the restatement above on other inputs"""
    lib.acehip_rt_debug_chacha20_block.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.acehip_rt_debug_chacha20_block.restype = None

    def block(key, counter, nonce):
        out = (C.c_uint32 * 16)()
        lib.acehip_rt_debug_chacha20_block((C.c_uint32 * 8)(*key), counter, (C.c_uint32 * 3)(*nonce), out)
        return list(out)

    key = [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)]
    got = block(key, 1, [0x09000000, 0x4A000000, 0])
    assert got == [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
                   0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    rng = np.random.default_rng(5)
    for _ in range(5):
        k = [int(v) for v in rng.integers(0, 1 << 32, 8)]
        n = [int(v) for v in rng.integers(0, 1 << 32, 3)]
        c = int(rng.integers(0, 1 << 32))
        assert block(k, c, n) == _chacha_block_py(k, c, n)
