"""Key generation and encryption of the product, on the GPU (-m gpu).

Keys are random by construction, so they cannot be compared with the reference's; what CAN be checked:

  1. structure -- read back from the key container (Acehip_rt_save_keys, "ACEHKEY1") and taken apart with the CPU oracle:
     the secret key is ternary with exactly `hamming_weight` non-zeros and balanced signs (Sample_ternary random_sample.c:99-150);
     pk0 + pk1*s, and b_j + a_j*s_old - P*s_new[digit j] for the relinearisation key and every rotation key
     (Generate_public_key ckks_key_generator.c:85-125, Generate_switching_key :127-200, Generate_rot_key :238-266) are ONE small
     polynomial on every limb, with the triangle distribution of Sample_triangle (random_sample.c:78-97: 0 w.p. 1/2, +-1 w.p. 1/4);
     a fresh encryption of zero (Encrypt_msg ckks_encryptor.c:20-95) decrypts to noise of the predicted variance;
  2. the reverse interop direction -- oracle/_ref/ct_parity_ref "load": the REFERENCE rtlib takes our keys and our ciphertexts
     (tests/c/ct_parity.c), reruns the operator script (HAdd, plaintext ops, Rescale, tensor product, Relinearize, Mul, Rotate at two
     levels, ModSwitch) and must reproduce every one of our results byte for byte, and decrypt our ciphertext to the message.
"""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu

REF_EXE = os.path.join(ROOT, "oracle", "_ref", "ct_parity_ref")

# "N mul_depth q0 sf dnum hamming slots level_after rot..." (level_after 0: no bootstrap -- the reference would have to set one up)
CONFIGS = {
    "n64_ternary": "64 6 60 50 3 0 32 0 1 -3 5",             # hamming weight 0: uniform ternary secret
    "n4096_hw192": "4096 8 51 50 3 192 2048 0 1 -5 64",      # the generated models' secret key distribution, ResNet-20 prime sizes
    "n65536_hw192": "65536 5 51 50 3 192 32768 0 1 -4096",   # the benchmark ring: N = 2^16 kernels, device sampler at full size
}


@pytest.fixture(scope="module")
def product_exe(tmp_path_factory):
    import ace_compiler_amd  # noqa: F401

    bmod = sys.modules["ace_compiler_amd.build"]
    bmod.build_rt()
    exe = str(tmp_path_factory.mktemp("ctp") / "ct_parity")
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-O1", "-w", os.path.join(ROOT, "tests", "c", "ct_parity.c"), "-I", inc, "-I", os.path.join(inc, "rt_ant"),
                           "-L", bmod.LIBDIR, "-lFHErt_ant", "-lFHErt_common", "-lm", "-Wl,-rpath," + bmod.LIBDIR, "-o", exe])
    return exe


@pytest.fixture(scope="module")
def made(product_exe, tmp_path_factory):
    """our library's keys, ciphertexts and results for every configuration (one GPU run each)"""
    out = {}
    for name, args in CONFIGS.items():
        d = str(tmp_path_factory.mktemp(name))
        env = dict(os.environ, ACEHIP_SEED="424242")
        r = subprocess.run([product_exe, "make", d] + args.split(), capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "made: keys.bin" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        assert oct(os.stat(os.path.join(d, "keys.bin")).st_mode & 0o777) == "0o600"  # key material is never group / world readable
        out[name] = d
    return out


def _read_keys(path):
    buf = open(path, "rb").read()
    assert buf[:8] == b"ACEHKEY1"
    ver, n, l, k, dnum, n_rot, n_auto, flags = struct.unpack_from("<8I", buf, 8)
    assert ver == 1 and flags == 0
    t = l + k
    off = 40
    primes = list(struct.unpack_from("<%dQ" % t, buf, off))
    off += 8 * t

    def take(limbs):
        nonlocal off
        a = np.frombuffer(buf, dtype=np.uint64, count=limbs * n, offset=off).reshape(limbs, n).copy()
        off += limbs * n * 8
        return a

    sk = take(t)
    pk0, pk1 = take(l), take(l)
    relin = np.stack([take(t) for _ in range(2 * dnum)]).reshape(dnum, 2, t, n)
    rots = [struct.unpack_from("<iI", buf, off + 8 * i) for i in range(n_rot)]
    off += 8 * n_rot
    autos = {}
    for _ in range(n_auto):
        idx = struct.unpack_from("<I", buf, off)[0]
        off += 8
        autos[idx] = np.stack([take(t) for _ in range(2 * dnum)]).reshape(dnum, 2, t, n)
    assert off == len(buf)
    return dict(N=n, L=l, K=k, dnum=dnum, primes=primes, sk=sk, pk0=pk0, pk1=pk1, relin=relin, rots=rots, autos=autos)


def _read_ct(path):
    buf = open(path, "rb").read()
    assert buf[:8] == b"ACEHCT01"
    n_polys, n, level, num_p, is_ntt, slots, sf_degree, _ = struct.unpack_from("<8I", buf, 8)
    polys = np.frombuffer(buf, dtype=np.uint64, offset=48).reshape(n_polys, level + num_p, n).copy()
    return polys, level, bool(is_ntt)


def _centered(o, x_ntt, gis):
    """coefficient-domain centred representatives, limb by limb: int64 [limbs][N]"""
    c = o.ntt_inv(x_ntt, gis)
    out = np.empty(c.shape, dtype=np.int64)
    for l, gi in enumerate(gis):
        q = np.uint64(o.primes[gi])
        v = c[l]
        out[l] = np.where(v > q // np.uint64(2), (v - q).astype(np.int64), v.astype(np.int64))
    return out


def _one_small_poly(o, x_ntt, gis, what):
    """x is the SAME small integer polynomial on every limb: returns it"""
    c = _centered(o, x_ntt, gis)
    for l in range(1, len(gis)):
        assert np.array_equal(c[l], c[0]), "%s: limb %d holds another polynomial than limb 0" % (what, l)
    return c[0]


def _is_triangle(e, what):
    n = e.size
    assert set(np.unique(e).tolist()) <= {-1, 0, 1}, "%s: values %s" % (what, np.unique(e)[:8])
    frac = np.count_nonzero(e) / n
    tol = 5 * (0.25 / n) ** 0.5 + 1e-9
    assert abs(frac - 0.5) <= tol, "%s: %.4f of the coefficients are non-zero, Sample_triangle gives 0.5 +- %.4f" % (what, frac, tol)
    plus = np.count_nonzero(e == 1) / max(1, np.count_nonzero(e))
    assert abs(plus - 0.5) <= 5 * (0.25 / max(1, np.count_nonzero(e))) ** 0.5 + 1e-9, "%s: signs are not balanced (%.3f)" % (what, plus)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_generated_keys_and_encryption_have_the_reference_structure(name, made):
    a = CONFIGS[name].split()
    n, depth, q0, sf, dnum_req, hamming = (int(x) for x in a[:6])
    kd = _read_keys(os.path.join(made[name], "keys.bin"))
    o = O.Oracle(n, depth + 1, q0, sf, dnum_req)
    try:
        L, K, T, dnum, alpha = o.L, o.K, o.L + o.K, o.dnum, o.alpha
        assert (kd["N"], kd["L"], kd["K"], kd["dnum"], kd["primes"]) == (n, L, K, dnum, o.primes)
        all_gi, q_gi = list(range(T)), list(range(L))
        # (i) the secret key: ternary, exact Hamming weight, balanced signs (random_sample.c:99-150)
        s = _one_small_poly(o, kd["sk"], all_gi, "secret key")
        assert set(np.unique(s).tolist()) <= {-1, 0, 1}
        if hamming:
            assert np.count_nonzero(s) == min(hamming, n)
            assert abs(int(np.count_nonzero(s == 1)) - hamming // 2) <= 1
        else:
            assert n // 2 <= np.count_nonzero(s) <= n  # uniform over {-1, 0, 1}: about 2/3
        # (ii) public key: pk0 + pk1*s = e (ckks_key_generator.c:85-125)
        e = _one_small_poly(o, o.hw_modadd(kd["pk0"], o.hw_modmul(kd["pk1"], kd["sk"][:L], q_gi), q_gi), q_gi, "public key")
        if n >= 1024:
            _is_triangle(e, "public key error")
        else:
            assert set(np.unique(e).tolist()) <= {-1, 0, 1}
        # switch keys: b_j + a_j*old - P*new on the limbs of digit j = e_j, ONE triangle polynomial on all L+K limbs (:127-200)
        p_mod = [int(np.prod([int(p) % int(o.primes[i]) for p in o.primes[L:]], dtype=object) % int(o.primes[i])) for i in range(T)]

        def check_switch_key(key, old, new, what):
            for j in range(dnum):
                r = o.hw_modadd(key[j][0], o.hw_modmul(key[j][1], old, all_gi), all_gi)
                for i in range(L):
                    if i // alpha == j:
                        qi = np.uint64(o.primes[i])
                        scaled = o.hw_modmul(new[i:i + 1], np.full((1, n), p_mod[i], dtype=np.uint64), [i])[0]
                        neg = np.where(scaled == 0, np.uint64(0), qi - scaled)
                        r[i] = o.hw_modadd(r[i:i + 1], neg[None, :], [i])[0]
                ej = _one_small_poly(o, r, all_gi, "%s, digit %d" % (what, j))
                if n >= 1024:
                    _is_triangle(ej, "%s error, digit %d" % (what, j))
                else:
                    assert set(np.unique(ej).tolist()) <= {-1, 0, 1}

        s2 = np.zeros((T, n), dtype=np.uint64)
        s2[:L] = o.hw_modmul(kd["sk"][:L], kd["sk"][:L], q_gi)   # relinearisation: new = s^2 on the q-limbs (:204-216)
        check_switch_key(kd["relin"], kd["sk"], s2, "relinearisation key")
        assert len(kd["rots"]) >= len(a) - 8 and len(kd["autos"]) >= 1
        for rot, k in kd["rots"][:3]:                             # rotation keys (:238-266): old = sigma_{k^-1}(s), new = s
            assert k == O.lib().orc_find_automorphism_index(rot, n)
            kinv = pow(k, -1, 2 * n)
            old = o.hw_rotate(kd["sk"], o.automorphism(kinv), all_gi)
            check_switch_key(kd["autos"][k], old, kd["sk"], "rotation key %d" % rot)
        # (iii) a fresh encryption of zero decrypts to noise v*e + e1 + e2*s: variance N/4 + 1/2 + h/2 (ckks_encryptor.c:20-95)
        ct, level, is_ntt = _read_ct(os.path.join(made[name], "zero.ct"))
        assert is_ntt and level == L
        noise = _one_small_poly(o, o.hw_modadd(ct[0], o.hw_modmul(ct[1], kd["sk"][:L], q_gi), q_gi), q_gi, "encryption of zero")
        h = np.count_nonzero(s)
        std = (n / 4 + 0.5 + h / 2) ** 0.5
        if n >= 1024:
            assert 0.85 * std <= noise.std() <= 1.15 * std, "noise std %.2f, predicted %.2f" % (noise.std(), std)
        assert np.abs(noise).max() <= 7 * std + 2
    finally:
        o.close()


@pytest.mark.parametrize("name", ["n64_ternary", "n4096_hw192"])
def test_reference_accepts_our_keys_and_ciphertexts(name, made):
    if not os.path.exists(REF_EXE):
        pytest.fail("oracle/_ref/ct_parity_ref not built (needs /root/reference: make -C oracle ref) -- a build output of the dev container that must travel with the snapshot")
    r = subprocess.run([REF_EXE, "load", made[name]] + CONFIGS[name].split(), capture_output=True, text=True, timeout=1800)
    tail = r.stdout[-6000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert "SUCESS! the reference reproduces every result" in r.stdout and "MISMATCH" not in r.stdout, tail
    assert "automorphism keys replaced" in r.stdout and "reference decrypts our in_a" in r.stdout
    for step in ("add", "sub", "add_plain", "plain", "mul_plain", "mul_plain_rescaled", "mul3", "relin", "relin_rescaled", "mul", "modswitch",
                 "rot_1", "rot_low_1"):
        assert "MATCH %s\n" % step in r.stdout, tail


def test_keyed_uniform_sampler_is_chacha20_reduced_mod_q():
    """acehip_sample_uniform_keyed (the device sampler behind the public `a` polynomials outside the ACEHIP_SEED test mode): coefficients
    4t..4t+3 of limb position pos = the four 128-bit quarters of the ChaCha20 block (key, counter t, nonce (pos, 'UNIF', 0)), top four bits
    dropped, reduced mod the limb's prime -- recomputed here with Python integers (RFC 8439 block function of tests/test_valid_helpers.py);
    and the samples look uniform (mean and variance of x / q over a limb)."""
    import ctypes as C

    import ace_compiler_amd as A
    from test_valid_helpers import _chacha_block_py

    N, L = 4096, 5
    rt = A.AceHip(N, L, 60, 50, 2, device=0)
    T = L + rt.K
    key = [0x03020100 + 0x04040404 * i for i in range(8)]
    d = rt.buf(T * N)
    rt.check(rt.lib.acehip_sample_uniform_keyed(rt.h, d.ptr, L, 0, T, (C.c_uint32 * 8)(*key), None))
    got = d.download((T, N))
    for pos in (0, 3, T - 1):
        q = int(rt.primes[pos])
        for t in (0, 1, 77, N // 4 - 1):
            w = _chacha_block_py(key, t, [pos, 0x554E4946, 0])
            for j in range(4):
                lo = w[4 * j] | (w[4 * j + 1] << 32)
                hi = w[4 * j + 2] | (w[4 * j + 3] << 32)
                assert int(got[pos, 4 * t + j]) == (((hi >> 4) << 64) | lo) % q, (pos, t, j)
        x = got[pos].astype(np.float64) / q
        assert abs(x.mean() - 0.5) < 0.02 and abs(x.var() - 1 / 12) < 0.01
    # another key, another stream; the same key, the same stream
    d2 = rt.buf(T * N)
    rt.check(rt.lib.acehip_sample_uniform_keyed(rt.h, d2.ptr, L, 0, T, (C.c_uint32 * 8)(*key), None))
    assert np.array_equal(d2.download((T, N)), got)
    key[0] ^= 1
    rt.check(rt.lib.acehip_sample_uniform_keyed(rt.h, d2.ptr, L, 0, T, (C.c_uint32 * 8)(*key), None))
    assert not np.array_equal(d2.download((T, N)), got)
    rt.close()


def test_unseeded_runs_use_fresh_randomness_and_still_decrypt(product_exe, tmp_path):
    """without ACEHIP_SEED every stream is ChaCha20 under 256 bits from getrandom(): two runs make different keys, and each run still
    decrypts its own ciphertexts to the messages (in_a through encrypt / decrypt, the product a * b through tensor product,
    relinearisation under the fresh key and rescale)"""
    digests = []
    for k in range(2):
        d = str(tmp_path / ("run%d" % k))
        os.makedirs(d)
        env = {k: v for k, v in os.environ.items() if k != "ACEHIP_SEED"}
        r = subprocess.run([product_exe, "make", d] + CONFIGS["n4096_hw192"].split(), capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0 and "made: keys.bin" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
        import hashlib
        import math
        import re

        vals = {m.group(1): [float(v) for v in m.group(2).split()] for m in re.finditer(r"msg_(\w+)\[0\.\.3\] =((?: -?\d+\.\d+)+)", r.stdout)}
        xa = [math.sin(0.37 * i) * 0.5 for i in range(4)]
        xb = [math.cos(0.23 * i + 1.0) * 0.4 for i in range(4)]
        assert max(abs(g - w) for g, w in zip(vals["in_a"], xa)) < 1e-6, vals["in_a"]
        assert max(abs(g - a * b) for g, a, b in zip(vals["relin_rescaled"], xa, xb)) < 1e-5, vals["relin_rescaled"]
        digests.append(hashlib.sha256(open(os.path.join(d, "keys.bin"), "rb").read()).hexdigest())
    assert digests[0] != digests[1]
