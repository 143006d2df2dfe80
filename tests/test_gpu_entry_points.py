"""Direct oracle tests (-m gpu) of the C-ABI entry points that round 1 only reached through example programs
(VERDICT r01 weak #2): acehip_key_inner_product, acehip_decomp, acehip_mod_up, acehip_mul_scalars, acehip_add_scalars,
acehip_values_to_rns, acehip_sample_uniform, and the round-2 acehip_bsgs_inner, acehip_key_inner_product_add.  Bit-exact against the oracle's building
blocks (oracle/ckks_oracle.c) composed as the reference composes them."""
import ctypes as C

import numpy as np
import pytest

import ace_compiler_amd as A
import _oracle as O

pytestmark = pytest.mark.gpu

# (the last set: six one-limb digits -- the any-number-of-digits form of the key inner product inside Mod_down's passes, 60-bit q0 and P)
SETS = [(16, 10, 60, 59, 3, 10), (64, 7, 60, 51, 3, 5), (4096, 6, 60, 50, 3, 6), (65536, 5, 51, 50, 2, 4), (65536, 6, 60, 50, 6, 6)]


@pytest.fixture(scope="module", params=SETS, ids=lambda s: "n%d_l%d_lv%d" % (s[0], s[1], s[5]))
def env(request):
    N, L, q0, sf, dnum, level = request.param
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    yield o, rt, level
    rt.close()
    o.close()


def _gis(o, level, n):
    return [o.gidx(l, level) for l in range(n)]


def _mulmod_rows(o, a, b, gis):
    return o.hw_modmul(a, b, gis)


def test_key_inner_product(env):
    """Fast_switch_key_ext ckks_evaluator.c:418-460: acc_k = sum_d key_k[d] (*) ext[d] over level+K limbs (no ModDown);
    the oracle composes it from Hw_modmul / Hw_modadd exactly like the generated loops (resnet20 .inc:7011-7036)."""
    o, rt, level = env
    N, K, T = o.N, o.K, o.L + o.K
    nd = o.num_decomp(level)
    E = level + K
    key = o.make_key(700)
    ext = np.stack([o.uniform(E, level, 710 + d) for d in range(nd)])
    gis = _gis(o, level, E)
    e0 = np.zeros((E, N), dtype=np.uint64)
    e1 = np.zeros((E, N), dtype=np.uint64)
    for d in range(nd):
        k0 = np.stack([key[d, 0, gi] for gi in gis])
        k1 = np.stack([key[d, 1, gi] for gi in gis])
        e0 = o.hw_modadd(e0, o.hw_modmul(k0, ext[d], gis), gis)
        e1 = o.hw_modadd(e1, o.hw_modmul(k1, ext[d], gis), gis)
    dk, de, a0, a1 = rt.to_device(key), rt.to_device(ext), rt.buf(E * N), rt.buf(E * N)
    rt.check(rt.lib.acehip_key_inner_product(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, level, None))
    assert np.array_equal(a0.download((E, N)), e0) and np.array_equal(a1.download((E, N)), e1)
    assert rt.lib.acehip_key_inner_product(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, o.L + 1, None) < 0
    for d in (dk, de, a0, a1):
        d.free()


def test_key_inner_product_add(env):
    """Fast_rotate_ext ckks_evaluator.c:539-575: the inner product with P * c0 added to the first accumulator on the q-limbs
    (acehip_key_inner_product_add); the oracle composes it as the reference does: Fast_switch_key_ext, then
    Scalars_integer_multiply_poly and Add_poly on the q part."""
    o, rt, level = env
    N, K = o.N, o.K
    nd = o.num_decomp(level)
    E = level + K
    key = o.make_key(730)
    ext = np.stack([o.uniform(E, level, 740 + d) for d in range(nd)])
    c0 = o.uniform(level, level, 750)
    gis = _gis(o, level, E)
    e0 = np.zeros((E, N), dtype=np.uint64)
    e1 = np.zeros((E, N), dtype=np.uint64)
    for d in range(nd):
        k0 = np.stack([key[d, 0, gi] for gi in gis])
        k1 = np.stack([key[d, 1, gi] for gi in gis])
        e0 = o.hw_modadd(e0, o.hw_modmul(k0, ext[d], gis), gis)
        e1 = o.hw_modadd(e1, o.hw_modmul(k1, ext[d], gis), gis)
    pm = []
    for i in range(level):                                   # P mod q_i
        r = 1
        for j in range(K):
            r = (r * (o.primes[o.L + j] % o.primes[i])) % o.primes[i]
        pm.append(r)
    scal = np.stack([np.full(N, w, dtype=np.uint64) for w in pm])
    qg = gis[:level]
    e0[:level] = o.hw_modadd(e0[:level], o.hw_modmul(c0, scal, qg), qg)
    dk, de, dc, a0, a1 = rt.to_device(key), rt.to_device(ext), rt.to_device(c0), rt.buf(E * N), rt.buf(E * N)
    hs = (C.c_uint64 * level)(*pm)
    rt.check(rt.lib.acehip_key_inner_product_add(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, level, dc.ptr, hs, None))
    assert np.array_equal(a0.download((E, N)), e0) and np.array_equal(a1.download((E, N)), e1)
    assert rt.lib.acehip_key_inner_product_add(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, level, None, hs, None) < 0
    bad = (C.c_uint64 * level)(*([o.primes[0]] + pm[1:]))   # not a residue of q_0
    assert rt.lib.acehip_key_inner_product_add(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, level, dc.ptr, bad, None) < 0
    for d in (dk, de, dc, a0, a1):
        d.free()


def test_rotate_add2(env):
    """res_z = acc_z + automorphism_k(a_z) on PQ-extended polynomials (Rotate_iteration's outer sums: Automorphism_transform
    then Add_poly, ckks_bootstrap_context.c:1343-1377), one and two polynomials, in place on the accumulator"""
    o, rt, level = env
    N, K = o.N, o.K
    E = level + K
    gis = _gis(o, level, E)
    k = rt.auto_index(-3)
    perm = np.asarray(o.automorphism(k, True), dtype=np.int64)
    acc = [o.uniform(E, level, 760 + z) for z in range(2)]
    a = [o.uniform(E, level, 770 + z) for z in range(2)]
    want = [o.hw_modadd(acc[z], np.ascontiguousarray(a[z][:, perm]), gis) for z in range(2)]
    d_acc = [rt.to_device(x) for x in acc]
    d_a = [rt.to_device(x) for x in a]
    d_r = rt.buf(E * N)
    # one polynomial, out of place
    rt.check(rt.lib.acehip_rotate_add2(rt.h, d_r.ptr, None, d_acc[0].ptr, None, d_a[0].ptr, None, k, level, 0, E, None))
    assert np.array_equal(d_r.download((E, N)), want[0])
    # two polynomials, in place on the accumulators
    rt.check(rt.lib.acehip_rotate_add2(rt.h, d_acc[0].ptr, d_acc[1].ptr, d_acc[0].ptr, d_acc[1].ptr, d_a[0].ptr, d_a[1].ptr, k, level, 0, E, None))
    assert np.array_equal(d_acc[0].download((E, N)), want[0]) and np.array_equal(d_acc[1].download((E, N)), want[1])
    # the rotated operand as a result, an even index
    assert rt.lib.acehip_rotate_add2(rt.h, d_a[0].ptr, None, d_acc[0].ptr, None, d_a[0].ptr, None, k, level, 0, E, None) < 0
    assert rt.lib.acehip_rotate_add2(rt.h, d_r.ptr, None, d_acc[0].ptr, None, d_a[0].ptr, None, 4, level, 0, E, None) < 0
    for d in d_acc + d_a + [d_r]:
        d.free()


def test_decomp_then_mod_up_equals_decomp_modup(env):
    """the unfused pair of the generated code (eg_fhertlib_relin.inc:79-80): Decomp copies the digit's limbs, Mod_up raises
    them; together they must give what Decomp_modup gives (oracle: Decompose_modup polynomial.c:1241-1335)."""
    o, rt, level = env
    N, K = o.N, o.K
    a = o.uniform(level, level, 720)
    da = rt.to_device(a)
    for d in range(o.num_decomp(level)):
        start = o.alpha * d
        n2 = min(o.alpha, level - start)
        dd, de = rt.buf(n2 * N), rt.buf((level + K) * N)
        assert rt.lib.acehip_decomp(rt.h, dd.ptr, da.ptr, level, d, None) == n2
        assert np.array_equal(dd.download((n2, N)), a[start:start + n2])
        assert rt.lib.acehip_mod_up(rt.h, de.ptr, dd.ptr, level, d, None) == n2
        assert np.array_equal(de.download((level + K, N)), o.decomp_modup(a, level, d)), d
        dd.free()
        de.free()
    assert rt.lib.acehip_decomp(rt.h, da.ptr, da.ptr, level, o.num_decomp(level), None) < 0
    da.free()


def test_mul_and_add_scalars(env):
    """Scalars_integer_multiply_poly polynomial.c:234-268 / Add_const ckks_evaluator.c:116-128: one scalar per limb,
    over q-limbs and p-limbs, sub-ranges included, scalars 0, 1, q-1 among them."""
    o, rt, level = env
    N, K = o.N, o.K
    E = level + K
    gis = _gis(o, level, E)
    a = o.uniform(E, level, 730)
    rng = np.random.default_rng(5)
    sc = np.array([int(rng.integers(0, o.primes[gi])) for gi in gis], dtype=np.uint64)
    sc[0], sc[-1] = 0, o.primes[gis[-1]] - 1
    if E > 2:
        sc[1] = 1
    da, dr = rt.to_device(a), rt.buf(E * N)
    srow = np.stack([np.full(N, s, dtype=np.uint64) for s in sc])
    rt.check(rt.lib.acehip_mul_scalars(rt.h, dr.ptr, da.ptr, sc.ctypes.data, level, 0, E, None))
    assert np.array_equal(dr.download((E, N)), o.hw_modmul(a, srow, gis))
    rt.check(rt.lib.acehip_add_scalars(rt.h, dr.ptr, da.ptr, sc.ctypes.data, level, 0, E, None))
    assert np.array_equal(dr.download((E, N)), o.hw_modadd(a, srow, gis))
    # the p-limbs alone (positions level .. level+K-1); pointers are polynomial bases, scalars start at position pos0
    rt.check(rt.lib.acehip_memset(dr.ptr, 0, E * N * 8, None))
    rt.check(rt.lib.acehip_mul_scalars(rt.h, dr.ptr, da.ptr, sc[level:].ctypes.data, level, level, K, None))
    got = dr.download((E, N))
    assert np.array_equal(got[level:], o.hw_modmul(a[level:], srow[level:], gis[level:])) and not got[:level].any()
    da.free()
    dr.free()


def test_values_to_rns(env):
    """Transform_values_to_rns polynomial.c:362-392: signed 64-bit values reduced into every limb, negative values and
    magnitudes far above the primes included."""
    o, rt, level = env
    N, K = o.N, o.K
    E = level + K
    rng = np.random.default_rng(6)
    v = rng.integers(-(1 << 62), 1 << 62, size=N, dtype=np.int64)
    v[:4] = [0, -1, 1, -(1 << 62)]
    dv, dr = rt.to_device(v.view(np.uint64)), rt.buf(E * N)
    rt.check(rt.lib.acehip_values_to_rns(rt.h, dr.ptr, dv.ptr, level, 0, E, None))
    got = dr.download((E, N))
    for l, gi in enumerate(_gis(o, level, E)):
        q = o.primes[gi]
        exp = np.array([int(x) % q for x in v.tolist()], dtype=np.uint64)
        assert np.array_equal(got[l], exp), l
    dv.free()
    dr.free()


def test_sample_uniform(env):
    """Sample_uniform_poly polynomial.c:1349-1371: residues in [0, q) on every limb, the same seed gives the same
    polynomial, different seeds and different limbs differ, and the values spread over the whole range (mean and
    top-bit frequency of a uniform variable within 6 sigma)."""
    o, rt, level = env
    N, K = o.N, o.K
    E = level + K
    d1, d2 = rt.buf(E * N), rt.buf(E * N)
    rt.check(rt.lib.acehip_sample_uniform(rt.h, d1.ptr, level, 0, E, 1234, None))
    rt.check(rt.lib.acehip_sample_uniform(rt.h, d2.ptr, level, 0, E, 1234, None))
    a, b = d1.download((E, N)), d2.download((E, N))
    assert np.array_equal(a, b)
    rt.check(rt.lib.acehip_sample_uniform(rt.h, d2.ptr, level, 0, E, 1235, None))
    c = d2.download((E, N))
    assert not np.array_equal(a, c)
    for l, gi in enumerate(_gis(o, level, E)):
        q = o.primes[gi]
        assert int(a[l].max()) < q
        if l:
            assert not np.array_equal(a[l], a[0])
        if N >= 4096:
            x = a[l].astype(np.float64) / q
            assert abs(x.mean() - 0.5) < 6 * (1 / 12 ** 0.5) / N ** 0.5
            assert abs((x >= 0.5).mean() - 0.5) < 6 * 0.5 / N ** 0.5
    d1.free()
    d2.free()


@pytest.mark.parametrize("g,b,missing", [(3, 2, None), (8, 8, 63), (13, 4, 51)])
def test_bsgs_inner(env, g, b, missing):
    """Rotate_iteration's inner loop (ckks_bootstrap_context.c:1326-1341): out_i = sum_j rot_j (*) pt_{i,j} on PQ-extended
    ciphertexts, with plaintexts that carry more q-limbs than the ciphertext level (Derive_plain) and one absent diagonal;
    oracle = Mul_plaintext + Add_ciphertext per term (Hw_modmul / Hw_modadd on every limb)."""
    o, rt, level = env
    N, K, L = o.N, o.K, o.L
    E = level + K
    gis = _gis(o, level, E)
    pt_q = L                                    # plaintexts encoded at the top level: L q-limbs then K p-limbs
    rot0 = [o.uniform(E, level, 800 + j) for j in range(g)]
    rot1 = [o.uniform(E, level, 830 + j) for j in range(g)]
    pts = {}
    for i in range(b):
        for j in range(g):
            if i * g + j != missing:
                pts[(i, j)] = o.uniform(L + K, L, 900 + i * g + j)
    exp0, exp1 = [], []
    for i in range(b):
        s0 = np.zeros((E, N), dtype=np.uint64)
        s1 = np.zeros((E, N), dtype=np.uint64)
        for j in range(g):
            if (i, j) in pts:
                p = np.concatenate([pts[(i, j)][:level], pts[(i, j)][L:]])   # the view Mul_plaintext uses at this level
                s0 = o.hw_modadd(s0, o.hw_modmul(rot0[j], p, gis), gis)
                s1 = o.hw_modadd(s1, o.hw_modmul(rot1[j], p, gis), gis)
        exp0.append(s0)
        exp1.append(s1)
    d_r0 = [rt.to_device(x) for x in rot0]
    d_r1 = [rt.to_device(x) for x in rot1]
    d_pt = {k: rt.to_device(v) for k, v in pts.items()}
    d_o0 = [rt.buf(E * N) for _ in range(b)]
    d_o1 = [rt.buf(E * N) for _ in range(b)]
    vp = C.c_void_p
    arr = lambda ptrs: (vp * len(ptrs))(*ptrs)  # noqa: E731
    pt_tab = arr([d_pt[(i, j)].ptr if (i, j) in d_pt else None for i in range(b) for j in range(g)])
    rt.check(rt.lib.acehip_bsgs_inner(rt.h, arr([d.ptr for d in d_o0]), arr([d.ptr for d in d_o1]), arr([d.ptr for d in d_r0]),
                                      arr([d.ptr for d in d_r1]), pt_tab, g, b, pt_q, level, None))
    for i in range(b):
        assert np.array_equal(d_o0[i].download((E, N)), exp0[i]), i
        assert np.array_equal(d_o1[i].download((E, N)), exp1[i]), i
    assert rt.lib.acehip_bsgs_inner(rt.h, arr([d.ptr for d in d_o0]), arr([d.ptr for d in d_o1]), arr([d.ptr for d in d_r0]),
                                    arr([d.ptr for d in d_r1]), pt_tab, 17, b, pt_q, level, None) < 0
    # the same products with the inputs given BEFORE their automorphisms (acehip_bsgs_inner_rot: Fast_rotate_ext's
    # Automorphism_transform applied where the kernel reads them): un-rotate the inputs on the host, pass the indices
    ks = [0 if j % 3 == 0 else rt.auto_index([1, -2, 5, 7][j % 4]) for j in range(g)]
    pre0, pre1 = [], []
    for j in range(g):
        if ks[j] == 0:
            pre0.append(rot0[j])
            pre1.append(rot1[j])
        else:
            perm = np.asarray(o.automorphism(ks[j], True), dtype=np.int64)   # rotated[i] = pre[perm[i]]
            inv = np.empty_like(perm)
            inv[perm] = np.arange(N)
            pre0.append(np.ascontiguousarray(rot0[j][:, inv]))
            pre1.append(np.ascontiguousarray(rot1[j][:, inv]))
            assert np.array_equal(pre0[j][:, perm], rot0[j])
    d_p0 = [rt.to_device(x) for x in pre0]
    d_p1 = [rt.to_device(x) for x in pre1]
    for d in d_o0 + d_o1:
        d.upload(np.zeros(E * N, dtype=np.uint64))
    autos = (C.c_uint32 * g)(*ks)
    rt.check(rt.lib.acehip_bsgs_inner_rot(rt.h, arr([d.ptr for d in d_o0]), arr([d.ptr for d in d_o1]), arr([d.ptr for d in d_p0]),
                                          arr([d.ptr for d in d_p1]), autos, pt_tab, g, b, pt_q, level, None))
    for i in range(b):
        assert np.array_equal(d_o0[i].download((E, N)), exp0[i]), i
        assert np.array_equal(d_o1[i].download((E, N)), exp1[i]), i
    even = (C.c_uint32 * g)(*([2] + [0] * (g - 1)))
    assert rt.lib.acehip_bsgs_inner_rot(rt.h, arr([d.ptr for d in d_o0]), arr([d.ptr for d in d_o1]), arr([d.ptr for d in d_p0]),
                                        arr([d.ptr for d in d_p1]), even, pt_tab, g, b, pt_q, level, None) < 0
    if g > 1 and ks[1] != 0:   # a rotated input that is also an output
        assert rt.lib.acehip_bsgs_inner_rot(rt.h, arr([d_p0[1].ptr] + [d.ptr for d in d_o0[1:]]), arr([d.ptr for d in d_o1]),
                                            arr([d.ptr for d in d_p0]), arr([d.ptr for d in d_p1]), autos, pt_tab, g, b, pt_q, level, None) < 0
    for d in d_r0 + d_r1 + d_o0 + d_o1 + d_p0 + d_p1 + list(d_pt.values()):
        d.free()


def test_reference_polynomial_kats_on_the_gpu():
    """rtlib/ant/unittest/ut_poly.cxx on the HIP path, N = 4, 10 primes, q0 = 60, Delta = 59, dnum = 3 (the parameters of its
    `precompute` tests): the negacyclic product {0,1,4,5} x {1,2,4,3} = {-29,-31,-9,17} (:178-216) through acehip_ntt_forward /
    acehip_modmul / acehip_ntt_inverse on every limb; rotation 3 of {0,1,4,59} = {0,-1,4,-59} and rotation 6 = identity (:112-143)
    through the NTT-domain automorphism; fused and unfused Switch_key_precompute agree (:349-381): acehip_modup_digits =
    acehip_decomp_modup per digit = acehip_decomp + acehip_mod_up."""
    N, L, q0, sf, dnum = 4, 10, 60, 59, 3
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    try:
        K, level = o.K, L
        primes = o.primes[:level]

        def rns(vals):
            return np.array([[v % q for v in vals] for q in primes], dtype=np.uint64)

        def centred(x):
            return [[int(v) - q if int(v) > q // 2 else int(v) for v in x[i]] for i, q in enumerate(primes)]

        fa, fb = rt.ntt(rns([0, 1, 4, 5]), level), rt.ntt(rns([1, 2, 4, 3]), level)
        prod = rt.ntt(rt.ew("modmul", fa, fb, level), level, inverse=True)
        assert centred(prod) == [[-29, -31, -9, 17]] * level
        assert centred(rt.ntt(rt.ew("modmul", fb, fa, level), level, inverse=True)) == [[-29, -31, -9, 17]] * level
        c = rt.ntt(rns([0, 1, 4, 59]), level)
        assert centred(rt.ntt(rt.rotate(c, rt.auto_index(3), level), level, inverse=True)) == [[0, -1, 4, -59]] * level
        assert centred(rt.ntt(rt.rotate(c, rt.auto_index(6), level), level, inverse=True)) == [[0, 1, 4, 59]] * level
        # Switch_key_precompute, fused and unfused
        x = fa
        nd = rt.num_decomp(level)
        E = level + K
        dx, dall = rt.to_device(x), rt.buf(nd * E * N)
        rt.check(rt.lib.acehip_modup_digits(rt.h, dall.ptr, dx.ptr, level, None))
        all_digits = dall.download((nd, E, N))
        for d in range(nd):
            fused = rt.decomp_modup(x, level, d)
            assert np.array_equal(fused, o.decomp_modup(x, level, d)) and np.array_equal(all_digits[d], fused)
            ddig, dout = rt.buf(o.alpha * N), rt.buf(E * N)
            rt.check(rt.lib.acehip_memset(dout.ptr, 0, E * N * 8, None))
            n2 = rt.lib.acehip_decomp(rt.h, ddig.ptr, dx.ptr, level, d, None)
            assert n2 > 0 and rt.lib.acehip_mod_up(rt.h, dout.ptr, ddig.ptr, level, d, None) == n2
            assert np.array_equal(dout.download((E, N)), fused), "unfused Decomp + Mod_up differs from Decomp_modup for digit %d" % d
            ddig.free()
            dout.free()
        dx.free()
        dall.free()
    finally:
        rt.close()
        o.close()


def test_keymac_mod_down2_matches_inner_product_then_mod_down(env):
    """acehip_keymac_mod_down2 = Fast_switch_key_ext (ckks_evaluator.c:418-460) + Reduce_rns_base of both accumulators (polynomial.c:928-967),
    with and without the form that never stores the accumulators (ACEHIP_KMAC_FUSE 2 / 0; at N = 2^16 the sums are formed inside the first
    inverse pass of the P-limbs and inside the Mod_down tail).  The digits are separate blocks, as the rt_ant shim hands them over.  Oracle:
    the generated loops' Hw_modmul / Hw_modadd chains, then Mod_down."""
    o, rt, level = env
    N, K, T = o.N, o.K, o.L + o.K
    nd = o.num_decomp(level)
    E = level + K
    key = o.make_key(900)
    ext = [o.uniform(E, level, 910 + d) for d in range(nd)]
    gis = _gis(o, level, E)
    e0 = np.zeros((E, N), dtype=np.uint64)
    e1 = np.zeros((E, N), dtype=np.uint64)
    for d in range(nd):
        k0 = np.stack([key[d, 0, gi] for gi in gis])
        k1 = np.stack([key[d, 1, gi] for gi in gis])
        e0 = o.hw_modadd(e0, o.hw_modmul(k0, ext[d], gis), gis)
        e1 = o.hw_modadd(e1, o.hw_modmul(k1, ext[d], gis), gis)
    want0, want1 = o.mod_down(e0, level), o.mod_down(e1, level)
    dk = rt.to_device(key)
    pad = rt.buf(3 * N)  # (keeps the digit blocks apart: no common stride)
    de = [rt.to_device(x) for x in ext]
    r0, r1 = rt.buf(level * N), rt.buf(level * N)
    h_ext = (C.c_void_p * nd)(*[d.ptr for d in de])
    h_key = (C.c_void_p * nd)(*[dk.at(d * 2 * T * N) for d in range(nd)])
    old = rt.lib.acehip_debug_set_kmac_fuse(1)
    try:
        for mode in (0, 2, 1):
            rt.lib.acehip_debug_set_kmac_fuse(mode)
            # (one image: the fused form where the q-limb passes are wide anyway -- more than 16 limb rows for the two polynomials)
            assert rt.lib.acehip_keymac_fusable(rt.h, level, nd) == (1 if (N == 65536 and (mode == 2 or (mode == 1 and 2 * level > 16))) else 0)
            r0.upload(np.zeros(level * N, dtype=np.uint64))
            r1.upload(np.zeros(level * N, dtype=np.uint64))
            rt.check(rt.lib.acehip_keymac_mod_down2(rt.h, r0.ptr, r1.ptr, h_ext, h_key, nd, level, None))
            assert np.array_equal(r0.download((level, N)), want0), mode
            assert np.array_equal(r1.download((level, N)), want1), mode
    finally:
        rt.lib.acehip_debug_set_kmac_fuse(old)
    assert rt.lib.acehip_keymac_mod_down2(rt.h, r0.ptr, r0.ptr, h_ext, h_key, nd, level, None) < 0
    # no aliasing (include/acehip.h): the last pass reads the digits while it writes the outputs
    assert rt.lib.acehip_keymac_mod_down2(rt.h, r0.ptr, r0.at(N), h_ext, h_key, nd, level, None) < 0          # outputs overlap
    assert rt.lib.acehip_keymac_mod_down2(rt.h, de[0].ptr, r1.ptr, h_ext, h_key, nd, level, None) < 0         # out0 = a raised digit
    assert rt.lib.acehip_keymac_mod_down2(rt.h, r0.ptr, de[nd - 1].at(K * N), h_ext, h_key, nd, level, None) < 0  # out1 inside a digit
    assert rt.lib.acehip_keymac_mod_down2(rt.h, r0.ptr, r1.ptr, h_ext, h_key, 0, level, None) < 0
    assert rt.lib.acehip_keymac_mod_down2(rt.h, r0.ptr, r1.ptr, h_ext, h_key, nd, o.L + 1, None) < 0
    for d in [dk, pad, r0, r1] + de:
        d.free()


def test_key_switch_with_the_inner_product_inside_mod_down(env):
    """acehip_key_switch (Fast_switch_key ckks_evaluator.c:391-460) gives the oracle's result whether or not the accumulators are stored
    (ACEHIP_KMAC_FUSE 0 / 2): own-digit limbs come from the input polynomial, the others from the raised digits."""
    o, rt, level = env
    a = o.uniform(level, level, 930)
    key = o.make_key(940)
    w0, w1 = o.key_switch(a, key, level)
    old = rt.lib.acehip_debug_set_kmac_fuse(1)
    try:
        for mode in (0, 2):
            rt.lib.acehip_debug_set_kmac_fuse(mode)
            g0, g1 = rt.key_switch(a, key, level)
            assert np.array_equal(g0, w0) and np.array_equal(g1, w1), mode
            # in place (d_out0 or d_out1 = d_in, allowed by include/acehip.h): the fused last pass reads the input while it writes the
            # outputs, so an aliased call must take the pipeline with stored accumulators -- same bits
            N = o.N
            dk = rt.to_device(key)
            for which in (0, 1):
                da, other = rt.to_device(a), rt.buf(level * N)
                outs = (da.ptr, other.ptr) if which == 0 else (other.ptr, da.ptr)
                rt.check(rt.lib.acehip_key_switch(rt.h, outs[0], outs[1], da.ptr, dk.ptr, level, None))
                got = (da.download((level, N)), other.download((level, N)))
                if which == 1:
                    got = got[::-1]
                assert np.array_equal(got[0], w0) and np.array_equal(got[1], w1), (mode, which)
                da.free()
                other.free()
            dk.free()
    finally:
        rt.lib.acehip_debug_set_kmac_fuse(old)


def test_key_inner_products_over_the_same_digits(env):
    """acehip_key_inner_products: the hoisted rotations of Rotate_iteration (ckks_bootstrap_context.c:1276-1290) -- several rotation keys over ONE
    set of raised digits, with and without the P * c0 addend of Fast_rotate_ext -- give what one acehip_key_inner_product[_add] call per key
    gives (those are checked against the oracle above); 1, 3 and 17 keys (more than one launch holds)."""
    o, rt, level = env
    N, K = o.N, o.K
    nd = o.num_decomp(level)
    E = level + K
    ext = np.stack([o.uniform(E, level, 760 + d) for d in range(nd)])
    c0 = o.uniform(level, level, 770)
    pm = []
    for i in range(level):
        r = 1
        for j in range(K):
            r = (r * (o.primes[o.L + j] % o.primes[i])) % o.primes[i]
        pm.append(r)
    hs = (C.c_uint64 * level)(*pm)
    de, dc = rt.to_device(ext), rt.to_device(c0)
    for n_keys in (1, 3, 17):
        dks = [rt.to_device(o.make_key(800 + 10 * j)) for j in range(n_keys)]
        for with_add in (False, True):
            want = []
            a0, a1 = rt.buf(E * N), rt.buf(E * N)
            for dk in dks:
                if with_add:
                    rt.check(rt.lib.acehip_key_inner_product_add(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, level, dc.ptr, hs, None))
                else:
                    rt.check(rt.lib.acehip_key_inner_product(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, level, None))
                want.append((a0.download((E, N)), a1.download((E, N))))
            outs = [(rt.buf(E * N), rt.buf(E * N)) for _ in range(n_keys)]
            h0 = (C.c_void_p * n_keys)(*[x[0].ptr for x in outs])
            h1 = (C.c_void_p * n_keys)(*[x[1].ptr for x in outs])
            hk = (C.c_void_p * n_keys)(*[dk.ptr for dk in dks])
            rt.check(rt.lib.acehip_key_inner_products(rt.h, h0, h1, hk, n_keys, de.ptr, level, dc.ptr if with_add else None,
                                                      hs if with_add else None, None))
            for j in range(n_keys):
                assert np.array_equal(outs[j][0].download((E, N)), want[j][0]), (n_keys, with_add, j)
                assert np.array_equal(outs[j][1].download((E, N)), want[j][1]), (n_keys, with_add, j)
            assert rt.lib.acehip_key_inner_products(rt.h, h0, h1, hk, n_keys, de.ptr, level, dc.ptr, None, None) < 0
            for x in outs:
                x[0].free()
                x[1].free()
            a0.free()
            a1.free()
        for dk in dks:
            dk.free()
    de.free()
    dc.free()


@pytest.mark.parametrize("cfg", [(4096, 6, 60, 50, 3, 6), (65536, 5, 51, 50, 2, 4)], ids=["n4096", "n65536"])
def test_key_inner_products_with_key_sets_inside_the_replicated_arena(cfg):
    """acehip_key_inner_products walks several rotation keys over one set of raised digits and reads every key part ONCE for all images of a
    launch -- right for keys outside the replicated arena (shared by all images; every caller of the shim allocates them there).  A key set
    INSIDE the arena is a different key per replica: such launches must take the single-key kernels, which rebase the key per replica
    (round-5 review: this used to give wrong sums silently).  Two replicas, three rotations, every block (digits, keys, outputs) inside the
    arena with different contents per replica; expected = the no-arena calls on plain buffers."""
    from ace_compiler_amd.binding import ArenaCfg

    N, L, q0, sf, dnum, level = cfg
    o = O.Oracle(N, L, q0, sf, dnum)
    rt = A.AceHip(N, L, q0, sf, dnum)
    try:
        K = o.K
        nd = o.num_decomp(level)
        E, T = level + K, L + K
        n_keys, n_rep = 3, 2
        key_words, ext_words, acc_words = nd * 2 * T * N, nd * E * N, E * N
        ext = [np.stack([o.uniform(E, level, 1760 + 10 * r + d) for d in range(nd)]) for r in range(n_rep)]
        keys = [[o.make_key(1800 + 100 * r + 10 * j) for j in range(n_keys)] for r in range(n_rep)]
        # expected: plain buffers, one call per key (the single-key entry point is checked against the oracle in test_key_inner_product)
        want = []
        for r in range(n_rep):
            de = rt.to_device(ext[r])
            a0, a1 = rt.buf(acc_words), rt.buf(acc_words)
            row = []
            for j in range(n_keys):
                dk = rt.to_device(keys[r][j])
                rt.check(rt.lib.acehip_key_inner_product(rt.h, a0.ptr, a1.ptr, dk.ptr, de.ptr, level, None))
                row.append((a0.download((E, N)), a1.download((E, N))))
                dk.free()
            want.append(row)
            for d in (de, a0, a1):
                d.free()
        rt.lib.acehip_workspace_words.restype = C.c_size_t
        ws_words = rt.lib.acehip_workspace_words(rt.h)
        gran = lambda w: (w + 31) // 32 * 32  # noqa: E731
        off, cur = {}, 0
        for name, words in [("ws", ws_words), ("sc", 2 * N), ("ext", ext_words)] + [("key%d" % j, key_words) for j in range(n_keys)] + \
                [("a0_%d" % j, acc_words) for j in range(n_keys)] + [("a1_%d" % j, acc_words) for j in range(n_keys)]:
            off[name] = cur
            cur += gran(words)
        rep_words = cur
        arena = rt.buf(rep_words * n_rep)
        for r in range(n_rep):
            base = r * rep_words
            rt.check(rt.lib.acehip_memcpy_h2d(arena.at(base + off["ext"]), np.ascontiguousarray(ext[r]).ctypes.data, ext_words * 8, None))
            for j in range(n_keys):
                kk = np.ascontiguousarray(keys[r][j])
                rt.check(rt.lib.acehip_memcpy_h2d(arena.at(base + off["key%d" % j]), kk.ctypes.data, key_words * 8, None))
        acfg = ArenaCfg(arena.ptr, rep_words * 8, rep_words * 8, n_rep, arena.at(off["ws"]), arena.at(off["sc"]), 2)
        rt.check(rt.lib.acehip_ctx_set_arena(rt.h, C.byref(acfg)))
        rt.check(rt.lib.acehip_ctx_select(rt.h, 0, n_rep))
        h0 = (C.c_void_p * n_keys)(*[arena.at(off["a0_%d" % j]) for j in range(n_keys)])
        h1 = (C.c_void_p * n_keys)(*[arena.at(off["a1_%d" % j]) for j in range(n_keys)])
        hk = (C.c_void_p * n_keys)(*[arena.at(off["key%d" % j]) for j in range(n_keys)])
        rt.check(rt.lib.acehip_key_inner_products(rt.h, h0, h1, hk, n_keys, arena.at(off["ext"]), level, None, None, None))
        got = arena.download((n_rep, rep_words))
        for r in range(n_rep):
            for j in range(n_keys):
                g0 = got[r, off["a0_%d" % j]:off["a0_%d" % j] + acc_words].reshape(E, N)
                g1 = got[r, off["a1_%d" % j]:off["a1_%d" % j] + acc_words].reshape(E, N)
                assert np.array_equal(g0, want[r][j][0]), (r, j)
                assert np.array_equal(g1, want[r][j][1]), (r, j)
        rt.check(rt.lib.acehip_ctx_select(rt.h, 0, 1))
        arena.free()
    finally:
        rt.close()
        o.close()


def test_malloc_limbs_outside_sharded_mode_is_a_plain_allocation(env):
    """acehip_malloc_limbs (the memory of switch keys): on a context that is not a rank of limb-sharded execution the whole block is backed
    like any allocation, usable by every kernel, counted by acehip_limb_memory and released by acehip_free.  (The owner-only form of a
    sharded rank is exercised over two processes in tests/test_gpu_batch_shard.py.)"""
    o, rt, level = env
    N, T = o.N, o.L + o.K
    n = 2 * T
    gi = (C.c_uint32 * n)(*[k % T for k in range(n)])
    b0, a0 = C.c_uint64(0), C.c_uint64(0)
    rt.lib.acehip_limb_memory(C.byref(b0), C.byref(a0))
    p = rt.lib.acehip_malloc_limbs(rt.h, gi, n)
    assert p
    b1, a1 = C.c_uint64(0), C.c_uint64(0)
    rt.lib.acehip_limb_memory(C.byref(b1), C.byref(a1))
    assert b1.value - b0.value == n * N * 8 and a1.value - a0.value == n * N * 8
    x = o.uniform(T, o.L, 4242)
    rt.check(rt.lib.acehip_memcpy_h2d(p, x.ctypes.data, T * N * 8, None))
    rt.check(rt.lib.acehip_memcpy_d2d(p + T * N * 8, p, T * N * 8, None))
    back = np.empty_like(x)
    rt.check(rt.lib.acehip_memcpy_d2h(back.ctypes.data, p + T * N * 8, T * N * 8, None))
    assert np.array_equal(back, x)
    bad = (C.c_uint32 * 1)(T)  # a prime index outside the chain
    assert not rt.lib.acehip_malloc_limbs(rt.h, bad, 1)
    rt.check(rt.lib.acehip_free(p))
    rt.lib.acehip_limb_memory(C.byref(b1), C.byref(a1))
    assert b1.value == b0.value and a1.value == a0.value
