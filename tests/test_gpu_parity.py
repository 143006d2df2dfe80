"""GPU parity tests (-m gpu): every HIP entry point of include/acehip.h against the CPU oracle on the
same seeded inputs (bit-exact), against the reference-generated golden fixtures, and -- at the full
BASELINE sizes -- through size-independent properties (NTT round trip, linearity, ModDown(ModUp))."""
import glob
import json
import os

import numpy as np
import pytest

import ace_compiler_amd as A
import _oracle as O
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SETS = [  # N, L, q0, sf, dnum, levels
    (8, 4, 60, 56, 2, [4, 3, 1]),
    (16, 3, 60, 50, 2, [3, 2]),
    (16, 10, 60, 59, 3, [10, 5, 4]),
    (32, 5, 33, 30, 0, [5, 2]),
    (64, 7, 60, 51, 3, [7, 4]),
    (1024, 7, 60, 51, 3, [6]),
    (4096, 4, 60, 50, 2, [4]),
    (8192, 4, 60, 50, 2, [4, 3]),
    (16384, 4, 60, 50, 2, [4]),
    (64, 40, 60, 50, 2, [40, 25]),   # alpha = 20, K = 17: more source limbs than one register chunk of the base conversion
    # the matrix-core base conversion (keyswitch.hip base_conv_mfma_kernel: N a multiple of 1024, primes above 2^32, digits and K of
    # at most 16 limbs): two k-steps (alpha = 12, K = 13), output tiles of 16 with ragged ends, short last digits, one limb
    (1024, 24, 60, 50, 2, [24, 17, 9, 1]),
    (2048, 30, 60, 55, 3, [30, 23, 16, 10]),   # alpha = 10
    (1024, 9, 60, 50, 3, [9, 8, 4]),          # one k-step (alpha = 3)
    (1024, 5, 33, 30, 0, [5, 2]),            # primes below 2^32: not eligible, the multiply-add kernels at the same ring size
]


@pytest.fixture(scope="module", params=SETS, ids=lambda s: "n%d_l%d_d%d" % (s[0], s[1], s[4]))
def pair(request):
    N, L, q0, sf, dnum, levels = request.param
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    yield o, rt, levels
    rt.close()
    o.close()


def test_context_matches(pair):
    o, rt, _ = pair
    assert rt.primes == o.primes and (rt.K, rt.alpha, rt.dnum) == (o.K, o.alpha, o.dnum)


def test_ntt(pair):
    o, rt, levels = pair
    level = levels[0]
    x = o.uniform(level + o.K, level, 21)
    gis = [o.gidx(l, level) for l in range(level + o.K)]
    f = rt.ntt(x, level)
    assert np.array_equal(f, o.ntt_fwd(x, gis))
    assert np.array_equal(rt.ntt(x, level, inverse=True), o.ntt_inv(x, gis))
    assert np.array_equal(rt.ntt(f, level, inverse=True), x)
    # a sub-range of limbs (positions 1..) and the p-limbs alone
    if level > 1:
        assert np.array_equal(rt.ntt(x[1:3], level, pos0=1), o.ntt_fwd(x[1:3], gis[1:3]))
    assert np.array_equal(rt.ntt(x[level:], level, pos0=level), o.ntt_fwd(x[level:], gis[level:]))


def test_elementwise_and_rotate(pair):
    o, rt, levels = pair
    level = levels[0]
    n = level + o.K
    gis = [o.gidx(l, level) for l in range(n)]
    a, b, c = o.uniform(n, level, 31), o.uniform(n, level, 32), o.uniform(n, level, 33)
    add = o.hw_modadd(a, b, gis)
    mul = o.hw_modmul(a, b, gis)
    assert np.array_equal(rt.ew("modadd", a, b, level), add)
    assert np.array_equal(rt.ew("modmul", a, b, level), mul)
    assert np.array_equal(rt.ew("modmuladd", a, b, level, acc=c), o.hw_modadd(c, mul, gis))
    # modsub: (a+b)-b == a
    assert np.array_equal(rt.ew("modsub", add, b, level), a)
    # edge values: 0 and q-1
    e = np.zeros_like(a)
    for l, gi in enumerate(gis):
        e[l, ::2] = o.primes[gi] - 1
    assert np.array_equal(rt.ew("modmul", e, e, level), o.hw_modmul(e, e, gis))
    assert np.array_equal(rt.ew("modadd", e, e, level), o.hw_modadd(e, e, gis))
    for rot in (1, -1, 3, o.N // 4):
        k = rt.auto_index(rot)
        assert np.array_equal(rt.rotate(a, k, level), o.hw_rotate(a, o.automorphism(k), gis))
    k = 2 * o.N - 1
    assert np.array_equal(rt.rotate(a, k, level), o.hw_rotate(a, o.automorphism(k), gis))


def test_single_limb_hw_calls(pair):
    """acehip_hw_*: the exact per-limb call shape of the generated code (poly_arith.c:14-56)."""
    o, rt, levels = pair
    N = o.N
    for gi in (0, o.L - 1, o.L, o.L + o.K - 1):
        q = o.primes[gi]
        rng = np.random.default_rng(gi)
        a = rng.integers(0, q, size=(1, N), dtype=np.uint64)
        b = rng.integers(0, q, size=(1, N), dtype=np.uint64)
        da, db, dr = rt.to_device(a), rt.to_device(b), rt.buf(N)
        rt.check(rt.lib.acehip_hw_modmul(rt.h, dr.ptr, da.ptr, db.ptr, gi, None))
        assert np.array_equal(dr.download((1, N)), o.hw_modmul(a, b, [gi]))
        rt.check(rt.lib.acehip_hw_modadd(rt.h, dr.ptr, da.ptr, db.ptr, gi, None))
        assert np.array_equal(dr.download((1, N)), o.hw_modadd(a, b, [gi]))
        k = rt.auto_index(1)
        rt.check(rt.lib.acehip_hw_rotate(rt.h, dr.ptr, da.ptr, rt.lib.acehip_auto_order(rt.h, k), gi, None))
        assert np.array_equal(dr.download((1, N)), o.hw_rotate(a, o.automorphism(k), [gi]))
        for d in (da, db, dr):
            d.free()


def test_decomp_modup_moddown_rescale(pair):
    o, rt, levels = pair
    for level in levels:
        a = o.uniform(level, level, 41 + level)
        for d in range(o.num_decomp(level)):
            assert np.array_equal(rt.decomp_modup(a, level, d), o.decomp_modup(a, level, d)), (level, d)
        x = o.uniform(level + o.K, level, 51 + level)
        assert np.array_equal(rt.mod_down(x, level), o.mod_down(x, level)), level
        if level > 1:
            assert np.array_equal(rt.rescale(a, level), o.rescale(a, level)), level


def test_base_conversion_extreme_values():
    """ModUp / ModDown on inputs that drive every sum of the base conversion to its largest and smallest value: all residues
    q_i - 1, all zero, and alternating -- the matrix-core form adds byte products of offset operands (keyswitch.hip
    base_conv_mfma_kernel: accumulator offsets, 80-bit recombination), so its range argument is checked at the ends."""
    N, L, q0, sf, dnum = 1024, 24, 60, 50, 2
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    try:
        for level in (24, 17, 5):
            T = level + o.K
            gis = [o.gidx(l, level) for l in range(T)]
            top = np.stack([np.full(N, o.primes[g] - 1, dtype=np.uint64) for g in gis])
            alt = top.copy()
            alt[:, ::2] = 0
            K, nd = o.K, o.num_decomp(level)
            for name, x in (("max", top), ("zero", np.zeros_like(top)), ("alternating", alt)):
                a = np.ascontiguousarray(x[:level])
                # the batched forms are the ones that take the matrix-core kernel (all digits at once; both polynomials of a pair)
                da, de = rt.to_device(a), rt.buf(nd * T * N)
                rt.check(rt.lib.acehip_modup_digits(rt.h, de.ptr, da.ptr, level, None))
                ext = de.download((nd, T, N))
                for d in range(nd):
                    assert np.array_equal(ext[d], o.decomp_modup(a, level, d)), (name, level, d)
                d0, d1, r0, r1 = rt.to_device(x), rt.to_device(x[:, ::-1].copy()), rt.buf(level * N), rt.buf(level * N)
                rt.check(rt.lib.acehip_mod_down2(rt.h, r0.ptr, r1.ptr, d0.ptr, d1.ptr, level, None))
                assert np.array_equal(r0.download((level, N)), o.mod_down(x, level)), (name, level)
                assert np.array_equal(r1.download((level, N)), o.mod_down(x[:, ::-1].copy(), level)), (name, level)
                for b in (da, de, d0, d1, r0, r1):
                    b.free()
    finally:
        rt.close()
        o.close()


def test_pair_forms_and_all_digit_modup(pair):
    """acehip_mod_down2 / acehip_rescale2 (c0 and c1 of a ciphertext in the same launches) and acehip_modup_digits
    (every digit at once) against the single-polynomial oracle results"""
    o, rt, levels = pair
    N, K = o.N, o.K
    for level in levels:
        x0, x1 = o.uniform(level + K, level, 71 + level), o.uniform(level + K, level, 72 + level)
        d0, d1, r0, r1 = rt.to_device(x0), rt.to_device(x1), rt.buf(level * N), rt.buf(level * N)
        rt.check(rt.lib.acehip_mod_down2(rt.h, r0.ptr, r1.ptr, d0.ptr, d1.ptr, level, None))
        assert np.array_equal(r0.download((level, N)), o.mod_down(x0, level)), level
        assert np.array_equal(r1.download((level, N)), o.mod_down(x1, level)), level
        assert np.array_equal(d0.download(x0.shape), x0) and np.array_equal(d1.download(x1.shape), x1)  # inputs intact
        assert rt.lib.acehip_mod_down2(rt.h, r0.ptr, r0.ptr, d0.ptr, d1.ptr, level, None) == -1
        for d in (d0, d1, r0, r1):
            d.free()
        a0, a1 = o.uniform(level, level, 73 + level), o.uniform(level, level, 74 + level)
        if level > 1:
            d0, d1, r0, r1 = rt.to_device(a0), rt.to_device(a1), rt.buf((level - 1) * N), rt.buf((level - 1) * N)
            rt.check(rt.lib.acehip_rescale2(rt.h, r0.ptr, r1.ptr, d0.ptr, d1.ptr, level, None))
            assert np.array_equal(r0.download((level - 1, N)), o.rescale(a0, level)), level
            assert np.array_equal(r1.download((level - 1, N)), o.rescale(a1, level)), level
            assert np.array_equal(d0.download(a0.shape), a0) and np.array_equal(d1.download(a1.shape), a1)
            for d in (d0, d1, r0, r1):
                d.free()
        nd = o.num_decomp(level)
        da, de = rt.to_device(a0), rt.buf(nd * (level + K) * N)
        rt.check(rt.lib.acehip_modup_digits(rt.h, de.ptr, da.ptr, level, None))
        ext = de.download((nd, level + K, N))
        for d in range(nd):
            assert np.array_equal(ext[d], o.decomp_modup(a0, level, d)), (level, d)
        da.free()
        de.free()


def test_key_switch(pair):
    o, rt, levels = pair
    key = o.make_key(1000)
    for level in levels:
        a = o.uniform(level, level, 61 + level)
        c0, c1 = rt.key_switch(a, key, level)
        e0, e1 = o.key_switch(a, key, level)
        assert np.array_equal(c0, e0) and np.array_equal(c1, e1), level


def test_argument_errors(pair):
    o, rt, levels = pair
    buf = rt.buf(o.N)
    assert rt.lib.acehip_ntt_forward(rt.h, buf.ptr, o.L + 1, 0, 1, None) == -1
    assert rt.lib.acehip_ntt_forward(rt.h, buf.ptr, o.L, o.L + o.K, 1, None) == -1
    assert rt.lib.acehip_rescale(rt.h, buf.ptr, buf.ptr, 1, None) == -1
    assert rt.lib.acehip_decomp_modup(rt.h, buf.ptr, buf.ptr, o.L, o.dnum, None) == -1
    assert rt.lib.acehip_hw_modmul(rt.h, buf.ptr, buf.ptr, buf.ptr, o.L + o.K, None) == -1
    buf.free()


# ---- golden fixtures produced by the reference itself ----
OPS_FILES = sorted(glob.glob(os.path.join(GOLDEN, "ref_ops_*.json")))


def _chk(vec, g, name):
    assert vec.size == g["n"], name
    if "data" in g:
        assert vec.reshape(-1).tolist() == g["data"], name
    assert O.sum64(vec) == g["sum64"] and O.xorw(vec) == g["xorw"], name


@pytest.mark.parametrize("path", OPS_FILES, ids=[os.path.basename(p)[8:-5] for p in OPS_FILES])
def test_against_reference_golden(path):
    g = json.load(open(path))
    N, L, K, level, seed = g["N"], g["L"], g["K"], g["level"], g["seed"]
    o = O.Oracle(N, L, g["q0_bits"], g["sf_bits"], g["dnum_req"])  # only used to rebuild the seeded inputs
    rt = A.AceHip(N, L, g["q0_bits"], g["sf_bits"], g["dnum_req"])
    try:
        a = o.uniform(level, level, seed)
        x = o.uniform(level + K, level, seed + 1)
        b = o.uniform(level, level, seed + 2)
        _chk(rt.ntt(x, level), g["ntt_fwd_x_ext"], "ntt_fwd")
        _chk(rt.ntt(x, level, inverse=True), g["ntt_inv_x_ext"], "ntt_inv")
        _chk(rt.mod_down(x, level), g["mod_down_x_ext"], "mod_down")
        _chk(rt.ew("modadd", a, b, level), g["hw_modadd_a_b"], "modadd")
        _chk(rt.ew("modmul", a, b, level), g["hw_modmul_a_b"], "modmul")
        for r in g["rotate"]:
            if r["rot_idx"] != 0:
                assert rt.auto_index(r["rot_idx"]) == r["k"]
            out = rt.rotate(a, r["k"], level)
            assert O.sum64(out) == r["out_sum64"] and O.xorw(out) == r["out_xorw"]
        for d in range(g["num_decomp"]):
            _chk(rt.decomp_modup(a, level, d), g["decomp_modup_%d" % d], "decomp_modup")
        if level > 1:
            _chk(rt.rescale(a, level), g["rescale_a"], "rescale")
        key = o.make_key(g["key_seed_base"])
        c0, c1 = rt.key_switch(a, key, level)
        _chk(c0, g["key_switch_c0"], "ks0")
        _chk(c1, g["key_switch_c1"], "ks1")
    finally:
        rt.close()
        o.close()


# ---- full BASELINE sizes: properties that need no oracle run ----
def test_full_size_properties():
    N, L, dnum = 65536, 25, 4
    rt = A.AceHip(N, L, 60, 56, dnum)
    o = O.Oracle(N, L, 60, 56, dnum)  # input generation + q values only
    try:
        level = L
        x = o.uniform(level + rt.K, level, 5)
        y = o.uniform(level + rt.K, level, 6)
        gis = [o.gidx(l, level) for l in range(level + rt.K)]
        fx = rt.ntt(x, level)
        assert np.array_equal(rt.ntt(fx, level, inverse=True), x)  # round trip
        # linearity: NTT(x + y) == NTT(x) + NTT(y)
        s = rt.ew("modadd", x, y, level)
        assert np.array_equal(rt.ntt(s, level), rt.ew("modadd", fx, rt.ntt(y, level), level))
        # convolution theorem on one limb: iNTT(NTT(x) * NTT(e1)) is the negacyclic shift of x by one
        e1 = np.zeros((1, N), dtype=np.uint64)
        e1[0, 1] = 1
        prod = rt.ew("modmul", fx[:1], rt.ntt(e1, level), level)
        sh = rt.ntt(prod, level, inverse=True)
        q0 = o.primes[0]
        exp = np.roll(x[:1], 1, axis=1)
        exp[0, 0] = (q0 - int(x[0, N - 1])) % q0
        assert np.array_equal(sh, exp)
        # ModDown(ModUp(a, digit)) == NTT(-v) with v the small overshoot of the uncorrected fast base
        # conversion: every coefficient of the centred iNTT is in [-alpha, 0]
        a = x[:level]
        ext = rt.decomp_modup(a, level, 0)
        dn = rt.mod_down(ext, level)
        co = rt.ntt(dn, level, inverse=True)
        alpha = rt.alpha
        for l in range(level):
            q = np.uint64(o.primes[l])
            v = co[l]
            neg = (q - v) % q
            assert int(neg.max()) <= alpha, (l, int(neg.max()))
    finally:
        rt.close()
        o.close()


def test_fused_ntt_paths_n65536():
    """N = 2^16 takes the fused transforms (ntt_fast.hip NttFuse: out-of-place first pass, Rescale / ModDown tail in the
    last pass): single and pair forms of Mod_down / Rescale, all-digit ModUp and the key-switch against the oracle."""
    N, L, q0, sf, dnum = 65536, 5, 60, 56, 2
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    try:
        K = o.K
        for level in (5, 3, 2):
            x0, x1 = o.uniform(level + K, level, 171 + level), o.uniform(level + K, level, 172 + level)
            e0, e1 = o.mod_down(x0, level), o.mod_down(x1, level)
            assert np.array_equal(rt.mod_down(x0, level), e0)
            d0, d1, r0, r1 = rt.to_device(x0), rt.to_device(x1), rt.buf(level * N), rt.buf(level * N)
            rt.check(rt.lib.acehip_mod_down2(rt.h, r0.ptr, r1.ptr, d0.ptr, d1.ptr, level, None))
            assert np.array_equal(r0.download((level, N)), e0) and np.array_equal(r1.download((level, N)), e1)
            assert np.array_equal(d0.download(x0.shape), x0) and np.array_equal(d1.download(x1.shape), x1)
            for d in (d0, d1, r0, r1):
                d.free()
            a0, a1 = o.uniform(level, level, 173 + level), o.uniform(level, level, 174 + level)
            f0, f1 = o.rescale(a0, level), o.rescale(a1, level)
            assert np.array_equal(rt.rescale(a0, level), f0)
            d0, d1, r0, r1 = rt.to_device(a0), rt.to_device(a1), rt.buf((level - 1) * N), rt.buf((level - 1) * N)
            rt.check(rt.lib.acehip_rescale2(rt.h, r0.ptr, r1.ptr, d0.ptr, d1.ptr, level, None))
            assert np.array_equal(r0.download((level - 1, N)), f0) and np.array_equal(r1.download((level - 1, N)), f1)
            assert np.array_equal(d0.download(a0.shape), a0) and np.array_equal(d1.download(a1.shape), a1)
            for d in (d0, d1, r0, r1):
                d.free()
            nd = o.num_decomp(level)
            da, de = rt.to_device(a0), rt.buf(nd * (level + K) * N)
            rt.check(rt.lib.acehip_modup_digits(rt.h, de.ptr, da.ptr, level, None))
            ext = de.download((nd, level + K, N))
            for d in range(nd):
                assert np.array_equal(ext[d], o.decomp_modup(a0, level, d)), (level, d)
            assert np.array_equal(da.download(a0.shape), a0)
            da.free()
            de.free()
            key = o.make_key(2000)
            c0, c1 = rt.key_switch(a0, key, level)
            g0, g1 = o.key_switch(a0, key, level)
            assert np.array_equal(c0, g0) and np.array_equal(c1, g1), level
    finally:
        rt.close()
        o.close()


@pytest.mark.parametrize("cfg", [(64, 7, 60, 51, 3, 6), (8192, 4, 60, 50, 2, 4), (65536, 5, 51, 50, 2, 4)], ids=["n64", "n8192", "n65536"])
def test_mod_raise(cfg):
    """acehip_mod_raise (bootstrap ModRaise, ckks_bootstrap_context.c:1527-1551) against the oracle: iNTT of limb 0, centred
    lift, reduction mod every prime of the raised level, NTT -- generic path (small N) and the fused N=2^16 path."""
    N, L, q0, sf, dnum, lv = cfg
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    try:
        a0, a1 = o.uniform(1, 1, 301), o.uniform(1, 1, 302)
        want = []
        for a in (a0, a1):
            coef = o.ntt_inv(a, [0])[0]
            q = np.uint64(o.primes[0])
            neg = coef > (q >> np.uint64(1))
            rows = np.empty((lv, N), dtype=np.uint64)
            for l in range(lv):
                ql = np.uint64(o.primes[l])
                pos = coef % ql
                rows[l] = np.where(neg, (ql - ((q - coef) % ql)) % ql, pos)
            want.append(o.ntt_fwd(rows, list(range(lv))))
        d0, d1, r = rt.to_device(a0), rt.to_device(a1), rt.buf(2 * lv * N)
        rt.check(rt.lib.acehip_mod_raise(rt.h, r.at(0), r.at(lv * N), d0.ptr, d1.ptr, lv, None))
        got = r.download((2, lv, N))
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        rt.check(rt.lib.acehip_mod_raise(rt.h, r.at(0), None, d1.ptr, None, lv, None))   # single polynomial form
        assert np.array_equal(r.download((2, lv, N))[0], want[1])
        for d in (d0, d1, r):
            d.free()
    finally:
        rt.close()
        o.close()


def test_conv_fusion_matches():
    """ACEHIP_CONV_FUSION=1 (base conversion computed by the first pass of the following NTT, ntt_fast.hip SRC_CONV*) is an
    opt-in variant of the N = 2^16 ModUp / ModDown pipelines: it must reproduce the reference-generated golden vectors bit for
    bit as well.  The switch is read once per process, so the golden tests run again in a child with the variable set."""
    import subprocess
    import sys

    env = dict(os.environ, ACEHIP_CONV_FUSION="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_against_reference_golden and n65536"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "deselected" in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("cfg", [(65536, 20, 51, 50, 3, (20, 13)), (65536, 12, 51, 48, 2, (12, 9))], ids=["delta50", "delta48"])
def test_fp_class_ntt_n65536(cfg):
    """The 48..50-bit scaling primes take FP64 butterflies in the wide N = 2^16 passes (csrc/ntt_fp.hpp): the same bits as the oracle
    (= the reference's canonical NTT, ntt.c:190-353) on random inputs, at the extreme values of every range argument of that arithmetic
    (all q-1, zero, alternating 0 / q-1, a single 1, a single q-1), through the plain transforms (limb 0, 51 bits, and the 60-bit
    P-limbs of the same launch stay on the integer classes) and through the fused neighbours: Mod_down / Rescale tails in the last
    pass, out-of-place first inverse pass, all-digit ModUp, key-switch."""
    N, L, q0, sf, dnum, levels = cfg
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    try:
        K = o.K
        level = levels[0]
        gis = [o.gidx(l, level) for l in range(level + K)]
        qs = np.array([o.primes[g] for g in gis], dtype=np.uint64)[:, None]
        x = o.uniform(level + K, level, 31)
        cases = {"uniform": x, "all_qm1": np.broadcast_to(qs - np.uint64(1), x.shape).copy(), "zero": np.zeros_like(x)}
        alt = np.zeros_like(x)
        alt[:, ::2] = (qs - np.uint64(1))
        cases["alternating"] = alt
        one = np.zeros_like(x)
        one[:, 1] = 1
        cases["single_one"] = one
        last = np.zeros_like(x)
        last[:, N - 1] = (qs - np.uint64(1))[:, 0]
        cases["single_qm1_last"] = last
        for name, v in cases.items():
            f = rt.ntt(v, level)
            assert np.array_equal(f, o.ntt_fwd(v, gis)), name
            assert np.array_equal(rt.ntt(v, level, inverse=True), o.ntt_inv(v, gis)), name
            assert np.array_equal(rt.ntt(f, level, inverse=True), v), name
        for level in levels:
            x0, x1 = o.uniform(level + K, level, 271 + level), o.uniform(level + K, level, 272 + level)
            e0, e1 = o.mod_down(x0, level), o.mod_down(x1, level)
            assert np.array_equal(rt.mod_down(x0, level), e0)
            d0, d1, r0, r1 = rt.to_device(x0), rt.to_device(x1), rt.buf(level * N), rt.buf(level * N)
            rt.check(rt.lib.acehip_mod_down2(rt.h, r0.ptr, r1.ptr, d0.ptr, d1.ptr, level, None))
            assert np.array_equal(r0.download((level, N)), e0) and np.array_equal(r1.download((level, N)), e1)
            for d in (d0, d1, r0, r1):
                d.free()
            a0, a1 = o.uniform(level, level, 273 + level), o.uniform(level, level, 274 + level)
            f0, f1 = o.rescale(a0, level), o.rescale(a1, level)
            assert np.array_equal(rt.rescale(a0, level), f0)
            d0, d1, r0, r1 = rt.to_device(a0), rt.to_device(a1), rt.buf((level - 1) * N), rt.buf((level - 1) * N)
            rt.check(rt.lib.acehip_rescale2(rt.h, r0.ptr, r1.ptr, d0.ptr, d1.ptr, level, None))
            assert np.array_equal(r0.download((level - 1, N)), f0) and np.array_equal(r1.download((level - 1, N)), f1)
            for d in (d0, d1, r0, r1):
                d.free()
            nd = o.num_decomp(level)
            da, de = rt.to_device(a0), rt.buf(nd * (level + K) * N)
            rt.check(rt.lib.acehip_modup_digits(rt.h, de.ptr, da.ptr, level, None))
            ext = de.download((nd, level + K, N))
            for d in range(nd):
                assert np.array_equal(ext[d], o.decomp_modup(a0, level, d)), (level, d)
            da.free()
            de.free()
            key = o.make_key(2100)
            c0, c1 = rt.key_switch(a0, key, level)
            g0, g1 = o.key_switch(a0, key, level)
            assert np.array_equal(c0, g0) and np.array_equal(c1, g1), level
    finally:
        rt.close()
        o.close()


@pytest.mark.parametrize("rows,tw8,fp,pipe", [("0", "65535", "1", "0"), ("100000", "65535", "1", "0"), ("0", "0", "1", "0"), ("0", "65535", "0", "0"),
                                              ("16", "65535", "0", "0"), ("0", "65535", "1", "2"), ("0", "0", "0", "2")],
                         ids=["wide_only", "narrow_always", "wide_16byte_twiddles", "wide_only_integer_classes", "default_widths_integer_classes",
                              "pipelined_two_tiles", "pipelined_two_tiles_integer_classes_16byte_twiddles"])
def test_ntt_tile_width_variants_match(rows, tw8, fp, pipe):
    """N = 2^16 transforms run as narrow passes (1024-coefficient tiles, ntt_fast.hip ntt4_*) up to ACEHIP_NTT_NARROW limb rows
    and as wide passes (4096-coefficient tiles) above: both must reproduce the reference-generated golden vectors and the
    fused-neighbour paths bit for bit whatever the size, so the N = 2^16 tests run again with each form forced -- once with
    the contiguous passes on the 16-byte twiddle tables (ACEHIP_NTT_TW8_POLYS=0) instead of the companion-only stream, and with the
    FP64 butterflies of the small primes switched off (ACEHIP_NTT_FP=0: every limb on the integer classes).  Round 6: the pipelined
    passes (ACEHIP_NTT_PIPE=2: a workgroup walks two tiles of a limb with the second tile's loads in flight during the first's
    butterflies; off by default -- measured slower, profiles/r06_ntt_pipelined_passes.txt -- but kept correct: same bits)."""
    import subprocess
    import sys

    env = dict(os.environ, ACEHIP_NTT_NARROW=rows, ACEHIP_NTT_TW8_POLYS=tw8, ACEHIP_NTT_FP=fp, ACEHIP_NTT_PIPE=pipe)
    tests = [os.path.abspath(__file__), os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_gpu_encode.py")]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu"] + tests + ["-k",
                        "(test_against_reference_golden and n65536) or test_fused_ntt_paths_n65536 or test_fp_class_ntt_n65536 or "
                        "(test_encode_matches_reference and n65536) or test_encode_batch_abi or (test_weight_prefetch and n65536)"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "deselected" in r.stdout, r.stdout[-500:]
