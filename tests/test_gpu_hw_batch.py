"""acehip_hw_batch (include/acehip.h): a list of per-limb Hw_modadd / Hw_modmul / Hw_rotate / copy / zero ops
must give exactly what issuing them one by one gives (the reference's call sequence, poly_arith.c:14-56), for any
mix of dependencies: accumulation chains, in-place operands, shared read-only operands, chains longer than one
launch, rotation runs with aliasing, partially overlapping limb pointers.  Checked bit-exactly against the oracle
applying the same program sequentially on the host."""
import numpy as np
import pytest

import _oracle as O

pytestmark = pytest.mark.gpu

N, L, Q0, SF, DNUM = 4096, 6, 60, 50, 3


@pytest.fixture(scope="module")
def env():
    import ace_compiler_amd as A
    from ace_compiler_amd import binding as B

    o = O.Oracle(N, L, Q0, SF, DNUM)
    rt = A.AceHip(N, L, Q0, SF, DNUM, device=0)
    yield o, rt, B
    rt.close()
    o.close()


def _arena(o, rows, seed):
    """rows x T limbs; column g holds residues of prime g"""
    T = o.L + o.K
    out = np.empty((rows * T, N), dtype=np.uint64)
    rng = np.random.default_rng(seed)
    for s in range(rows * T):
        out[s] = rng.integers(0, o.primes[s % T], size=N, dtype=np.uint64)
    return out


def _apply_host(o, B, arena, prog, perms):
    flat = arena.reshape(-1)

    def limb(off):
        return flat[off:off + N]

    for op, gi, r, a, b in prog:
        if op == B.HW_ADD:
            res = o.hw_modadd(limb(a).reshape(1, N).copy(), limb(b).reshape(1, N).copy(), [gi])[0]
        elif op == B.HW_MUL:
            res = o.hw_modmul(limb(a).reshape(1, N).copy(), limb(b).reshape(1, N).copy(), [gi])[0]
        elif op == B.HW_SUB:
            neg = (np.uint64(o.primes[gi]) - limb(b)) % np.uint64(o.primes[gi])
            res = o.hw_modadd(limb(a).reshape(1, N).copy(), neg.reshape(1, N).copy(), [gi])[0]
        elif op == B.HW_MULADD:
            t = o.hw_modmul(limb(a).reshape(1, N).copy(), limb(b).reshape(1, N).copy(), [gi])
            res = o.hw_modadd(limb(r).reshape(1, N).copy(), t, [gi])[0]
        elif op in (B.HW_MULC, B.HW_ADDC):
            k = np.full((1, N), b, dtype=np.uint64)   # b carries the residue itself
            fn = o.hw_modmul if op == B.HW_MULC else o.hw_modadd
            res = fn(limb(a).reshape(1, N).copy(), k, [gi])[0]
        elif op == B.HW_ROTATE:
            res = limb(a)[perms[b]].copy()
        elif op == B.HW_COPY:
            res = limb(a).copy()
        else:
            res = np.zeros(N, dtype=np.uint64)
        flat[r:r + N] = res


def _run_device(rt, B, arena, prog, perm_bufs, dead=None):
    """dead: [(word offset, words)] given up by the caller -> acehip_hw_batch_discard"""
    d = rt.to_device(arena)
    ops = []
    for op, gi, r, a, b in prog:
        if op == B.HW_ROTATE:
            bp = perm_bufs[b].ptr
        elif op in (B.HW_MULC, B.HW_ADDC):
            bp = b                      # immediate residue
        elif op in (B.HW_ADD, B.HW_MUL, B.HW_SUB, B.HW_MULADD):
            bp = d.at(b)
        else:
            bp = None
        ops.append((op, gi, d.at(r), d.at(a) if op != B.HW_ZERO else None, bp))
    if dead is None:
        rt.hw_batch(ops)
    else:
        rt.hw_batch_discard(ops, [(d.at(off), words) for off, words in dead])
    out = d.download(arena.shape)
    d.free()
    return out


def _random_program(o, B, rows, n_ops, seed, rot_keys):
    T = o.L + o.K
    rng = np.random.default_rng(seed)
    prog = []
    while len(prog) < n_ops:
        kind = rng.choice(["ew", "ew", "ew", "rot"]) if rot_keys else "ew"
        if kind == "ew":
            for _ in range(int(rng.integers(1, 40))):
                gi = int(rng.integers(0, T))
                r, a, b = (int((gi + T * rng.integers(0, rows)) * N) for _ in range(3))
                op = int(rng.choice([B.HW_ADD, B.HW_ADD, B.HW_MUL, B.HW_MUL, B.HW_COPY, B.HW_ZERO, B.HW_SUB, B.HW_MULADD,
                                     B.HW_MULADD, B.HW_MULC, B.HW_ADDC]))
                if op in (B.HW_MULC, B.HW_ADDC):
                    b = int(rng.integers(0, o.primes[gi]))
                prog.append((op, gi, r, a, b))
        else:
            k = int(rng.choice(rot_keys))
            for _ in range(int(rng.integers(1, 12))):
                gi = int(rng.integers(0, T))
                r, a = (int((gi + T * rng.integers(0, rows)) * N) for _ in range(2))
                if r != a:
                    prog.append((B.HW_ROTATE, gi, r, a, k))
    return prog[:n_ops]


def _check(env, prog, rows, seed, rot_keys=()):
    o, rt, B = env
    arena = _arena(o, rows, seed)
    perms = {k: np.asarray(o.automorphism(k, True), dtype=np.int64) for k in rot_keys}
    bufs = {}
    for k in rot_keys:
        bufs[k] = rt.buf(N, np.uint32).upload(perms[k].astype(np.uint32))
    want = arena.copy()
    _apply_host(o, B, want, prog, perms)
    got = _run_device(rt, B, arena, prog, bufs)
    for b in bufs.values():
        b.free()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_programs(env, seed):
    o, rt, B = env
    ks = [rt.auto_index(1), rt.auto_index(-3), rt.auto_index(64)]
    _check(env, _random_program(o, B, rows=3, n_ops=700, seed=seed, rot_keys=ks), rows=3, seed=10 + seed, rot_keys=ks)


def test_generated_conv_loop_shape(env):
    """the per-limb loop body of a generated conv (resnet20_cifar10_pre.onnx.inc:1492-1503): tmp = ct * pt on c0 and
    c1, acc += tmp -- 4 ops per limb, repeated for 9 kernel taps without a flush in between"""
    o, rt, B = env
    T = o.L + o.K

    def at(row, g):
        return (row * T + g) * N

    prog = []
    for tap in range(9):
        for g in range(o.L):
            prog += [(B.HW_MUL, g, at(4, g), at(0, g), at(2, g)), (B.HW_MUL, g, at(5, g), at(1, g), at(2, g)),
                     (B.HW_ADD, g, at(6, g), at(6, g), at(4, g)), (B.HW_ADD, g, at(7, g), at(7, g), at(5, g))]
    _check(env, prog, rows=8, seed=5)


def test_generated_key_inner_product_shape(env):
    """the key inner product loop of the generated Rotate() (resnet20_cifar10_pre.onnx.inc:7011-7036): every limb's
    product goes through ONE scratch limb (tmp[0]) before it is accumulated -- false dependencies that the batch
    breaks by renaming; the final content of the scratch limb must still be the last product"""
    o, rt, B = env
    T = o.L + o.K

    def at(row, g):
        return (row * T + g) * N

    tmp = at(9, 0)
    prog = []
    for g in range(T):
        prog += [(B.HW_MUL, g, tmp, at(2, g), at(0, g)), (B.HW_ADD, g, at(4, g), at(4, g), tmp),
                 (B.HW_MUL, g, tmp, at(3, g), at(0, g)), (B.HW_ADD, g, at(5, g), at(5, g), tmp)]
    _check(env, prog, rows=10, seed=12)
    # the same with zero fills of the accumulators in front and a reader of the scratch limb behind
    prog2 = [(B.HW_ZERO, 0, at(4, g), 0, 0) for g in range(T)] + [(B.HW_ZERO, 0, at(5, g), 0, 0) for g in range(T)] + prog
    prog2 += [(B.HW_COPY, 0, at(8, 0), tmp, 0), (B.HW_ADD, T - 1, at(7, T - 1), tmp, tmp)]
    _check(env, prog2, rows=10, seed=13)


def _check_discard(env, prog, rows, seed, dead_rows, rot_keys=()):
    """rows listed in dead_rows are handed over as given-up memory: every other limb must equal the sequential result"""
    o, rt, B = env
    T = o.L + o.K
    arena = _arena(o, rows, seed)
    perms = {k: np.asarray(o.automorphism(k, True), dtype=np.int64) for k in rot_keys}
    bufs = {k: rt.buf(N, np.uint32).upload(perms[k].astype(np.uint32)) for k in rot_keys}
    want = arena.copy()
    _apply_host(o, B, want, prog, perms)
    got = _run_device(rt, B, arena, prog, bufs, dead=[(r * T * N, T * N) for r in dead_rows])
    for b in bufs.values():
        b.free()
    keep = [s for s in range(rows * T) if s // T not in dead_rows]
    assert np.array_equal(got[keep], want[keep])
    return got, want


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_programs_with_given_up_rows(env, seed):
    o, rt, B = env
    ks = [rt.auto_index(1), rt.auto_index(-3)] if seed % 2 else []
    prog = _random_program(o, B, rows=5, n_ops=900, seed=40 + seed, rot_keys=ks)
    _check_discard(env, prog, rows=5, seed=50 + seed, dead_rows={1, 3} if seed < 3 else {0, 1, 2}, rot_keys=ks)


def test_generated_conv_loop_with_freed_temporaries(env):
    """test_generated_conv_loop_shape with the products' blocks (rows 4, 5) freed before the list is handed over, as the
    generated code does (Free_poly_data right behind the loop, resnet20_cifar10_pre.onnx.inc:1464-1471): the accumulators
    must come out the same; and the products must NOT have been written (the rows keep their old contents)"""
    o, rt, B = env
    T = o.L + o.K

    def at(row, g):
        return (row * T + g) * N

    prog = []
    for tap in range(9):
        for g in range(o.L):
            prog += [(B.HW_MUL, g, at(4, g), at(0, g), at(2, g)), (B.HW_MUL, g, at(5, g), at(1, g), at(2, g)),
                     (B.HW_ADD, g, at(6, g), at(6, g), at(4, g)), (B.HW_ADD, g, at(7, g), at(7, g), at(5, g))]
    got, want = _check_discard(env, prog, rows=8, seed=5, dead_rows={4, 5})
    before = _arena(o, 8, 5)
    assert np.array_equal(got[4 * T:4 * T + o.L], before[4 * T:4 * T + o.L])   # unspecified by contract; this is what is saved


def test_long_accumulation_chain(env):
    """one chain of 300 dependent ops (more than one launch can hold) on a single limb"""
    o, rt, B = env
    T = o.L + o.K
    g = 1
    acc, x, y = g * N, (T + g) * N, (2 * T + g) * N
    prog = []
    for i in range(150):
        prog += [(B.HW_MUL, g, x, x, y), (B.HW_ADD, g, acc, acc, x)]
    _check(env, prog, rows=3, seed=6)


def test_partially_overlapping_limbs_keep_call_order(env):
    """limb pointers that overlap by half a limb cannot be reordered: the result must still be the sequential one"""
    o, rt, B = env
    g = 0
    prog = [(B.HW_ADD, g, 0, 0, N // 2), (B.HW_COPY, g, N // 2, 0, 0), (B.HW_MUL, g, 2 * N, N // 2, 0),
            (B.HW_ZERO, g, N + N // 2, 0, 0), (B.HW_ADD, g, N, N // 2, 2 * N)]
    arena = np.empty((4, N), dtype=np.uint64)
    arena[:] = np.random.default_rng(3).integers(0, o.primes[0], size=(4, N), dtype=np.uint64)
    want = arena.copy()
    _apply_host(o, B, want, prog, {})
    got = _run_device(rt, B, arena, prog, {})
    assert np.array_equal(got, want)


def test_rotation_runs_with_aliasing(env):
    """a rotation whose source is the result of an earlier rotation of the same list, and results reused as sources"""
    o, rt, B = env
    T = o.L + o.K
    k = rt.auto_index(5)
    prog = []
    for g in range(T):
        a, b, c = g * N, (T + g) * N, (2 * T + g) * N
        prog += [(B.HW_ROTATE, g, b, a, k), (B.HW_ROTATE, g, c, b, k), (B.HW_ROTATE, g, a, c, k)]
    _check(env, prog, rows=3, seed=8, rot_keys=[k])


def test_bad_arguments_fail_loudly(env):
    o, rt, B = env
    import ace_compiler_amd as A

    d = rt.buf(2 * N)
    with pytest.raises(A.AceHipError, match="in-place rotation"):
        rt.hw_batch([(B.HW_ROTATE, 0, d.at(0), d.at(0), d.at(N))])
    with pytest.raises(A.AceHipError, match="prime index"):
        rt.hw_batch([(B.HW_ADD, 99, d.at(0), d.at(0), d.at(N))])
    with pytest.raises(A.AceHipError, match="not a residue"):
        rt.hw_batch([(B.HW_MULC, 0, d.at(0), d.at(0), rt.primes[0])])
    with pytest.raises(A.AceHipError, match="null operand"):
        rt.hw_batch([(B.HW_MUL, 0, d.at(0), d.at(0), None)])
    d.free()


@pytest.mark.parametrize("images,per_lane", [(2, "3"), (5, "3"), (5, "2"), (4, "4"), (3, "0")])
def test_random_programs_over_image_replicas(images, per_lane):
    """Image batches: one launch covers several replicas of the arena; operands outside the arena (weight plaintexts) are shared by all
    images.  Round 6: the per-limb kernel keeps several images of a batch in ONE lane (ACEHIP_HW_IMAGES_PER_LANE, default 3: a shared
    operand is loaded once for them; hw_batch.hip hw_batch_ew_im_kernel) -- every image must still get exactly what the sequential host
    program gives on ITS data, for batches that are not a multiple of the group (5 = 3 + 2, 2 < 3) and with the form off ("0").  The
    switch is read once per process: the case runs in a child."""
    import os
    import subprocess
    import sys

    code = r'''
import ctypes as C, numpy as np, sys
sys.path.insert(0, %r)
import _oracle as O, ace_compiler_amd as A
from ace_compiler_amd import binding as B
from ace_compiler_amd.binding import ArenaCfg
from test_gpu_hw_batch import _apply_host, N, L, Q0, SF, DNUM
R = %d
o = O.Oracle(N, L, Q0, SF, DNUM); rt = A.AceHip(N, L, Q0, SF, DNUM, device=0)
T = o.L + o.K; rows = 3
rng = np.random.default_rng(77)
rt.lib.acehip_workspace_words.restype = C.c_size_t
ws = (rt.lib.acehip_workspace_words(rt.h) + 31) // 32 * 32
off_sc, off_data = ws, ws + 2 * N
rep_words = off_data + rows * T * N
shared = np.stack([rng.integers(0, o.primes[g], size=N, dtype=np.uint64) for g in range(T)])
dsh = rt.to_device(shared)
data = [np.stack([rng.integers(0, o.primes[s %% T], size=N, dtype=np.uint64) for s in range(rows * T)]) for _ in range(R)]
arena = rt.buf(rep_words * R)
for r in range(R):
    rt.check(rt.lib.acehip_memcpy_h2d(arena.at(r * rep_words + off_data), np.ascontiguousarray(data[r]).ctypes.data, rows * T * N * 8, None))
cfg = ArenaCfg(arena.ptr, rep_words * 8, rep_words * 8, R, arena.at(0), arena.at(off_sc), 2)
rt.check(rt.lib.acehip_ctx_set_arena(rt.h, C.byref(cfg)))
rt.check(rt.lib.acehip_ctx_select(rt.h, 0, R))
# a program of multiply-accumulate chains against shared operands, plain ops between the image's own limbs, copies, immediates
prog = []
for _ in range(400):
    gi = int(rng.integers(0, T))
    r_, a_ = (int((gi + T * rng.integers(0, rows)) * N) for _ in range(2))
    op = int(rng.choice([B.HW_ADD, B.HW_MUL, B.HW_MUL, B.HW_MULADD, B.HW_MULADD, B.HW_SUB, B.HW_COPY, B.HW_MULC, B.HW_ZERO]))
    if op == B.HW_MULC:
        b_ = ("imm", int(rng.integers(0, o.primes[gi])))
    elif rng.random() < 0.5:
        b_ = ("shared", gi)
    else:
        b_ = ("own", int((gi + T * rng.integers(0, rows)) * N))
    prog.append((op, gi, r_, a_, b_))
ops = []
for op, gi, r_, a_, b_ in prog:
    bp = b_[1] if b_[0] == "imm" else (dsh.at(b_[1] * N) if b_[0] == "shared" else arena.at(off_data + b_[1]))
    if op in (B.HW_COPY, B.HW_ZERO):
        bp = None
    ops.append((op, gi, arena.at(off_data + r_), arena.at(off_data + a_) if op != B.HW_ZERO else None, bp))
rt.hw_batch(ops)
got = arena.download((R, rep_words))
for r in range(R):
    # host: the image's rows followed by the shared row block, so that every operand is an offset into one array
    both = np.concatenate([data[r], shared]).copy()
    hp = [(op, gi, r_, a_, (b_[1] if b_[0] == "imm" else (rows * T * N + b_[1] * N if b_[0] == "shared" else b_[1]))) for op, gi, r_, a_, b_ in prog]
    _apply_host(o, B, both, hp, {})
    assert np.array_equal(got[r, off_data:off_data + rows * T * N].reshape(rows * T, N), both[:rows * T]), "image %%d" %% r
    assert np.array_equal(both[rows * T:], shared)
print("replicas ok")
''' % (os.path.dirname(os.path.abspath(__file__)), images)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ACEHIP_HW_IMAGES_PER_LANE=per_lane), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "replicas ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
