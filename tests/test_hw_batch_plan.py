"""Host logic of acehip_hw_batch without a GPU: acehip_hw_batch_plan (include/acehip.h) returns what would be launched --
dead zero fills / copies dropped, intermediate versions renamed to scratch limbs, ops grouped into chain segments per
launch.  The plan is REPLAYED here on a numpy memory image: launches in order, the segments of a launch in a random order
(they run concurrently on the GPU), ops of a segment in order.  Whatever the order, the final contents of the caller's
memory must equal those of executing the original list one op at a time (the reference's call sequence,
poly_arith.c:14-56).  Exact modular arithmetic via Python integers on a tiny ring (N = 8)."""
import ctypes as C
import random

import numpy as np
import pytest

import ace_compiler_amd as A
from ace_compiler_amd import binding as B

N, L, Q0, SF, DNUM = 8, 4, 60, 56, 2
BASE = 0x10000000          # fake device address of the arena (never dereferenced: plan only)
SCRATCH = 0x7F0000000000   # fake scratch arena
SPAN = N * 8


@pytest.fixture(scope="module")
def rt():
    r = A.AceHip(N, L, Q0, SF, DNUM, host_only=True)
    yield r
    r.close()


def _exec(mem, primes, perms, op, gi, res, a, b):
    """one op on the address -> numpy row dictionary `mem` (object arrays of Python ints: exact)"""
    q = primes[gi] if op not in (B.HW_ROTATE, B.HW_COPY, B.HW_ZERO) else None
    if op == B.HW_ZERO:
        out = [0] * N
    elif op == B.HW_COPY:
        out = list(mem[a])
    elif op == B.HW_ROTATE:
        out = [mem[a][j] for j in perms[b]]
    elif op == B.HW_ADD:
        out = [(x + y) % q for x, y in zip(mem[a], mem[b])]
    elif op == B.HW_SUB:
        out = [(x - y) % q for x, y in zip(mem[a], mem[b])]
    elif op == B.HW_MUL:
        out = [(x * y) % q for x, y in zip(mem[a], mem[b])]
    elif op == B.HW_MULADD:
        out = [(r + x * y) % q for r, x, y in zip(mem[res], mem[a], mem[b])]
    elif op == B.HW_MULC:
        out = [(x * b) % q for x in mem[a]]
    else:  # ADDC
        out = [(x + b) % q for x in mem[a]]
    mem[res] = out


def _plan(rt, prog, dead=()):
    arr = (B.HwOp * len(prog))(*[B.HwOp(o, g, r, a or None, b or None) for o, g, r, a, b in prog])
    cap = 4 * len(prog) + 16
    out = (B.HwOp * cap)()
    launch = (C.c_uint32 * cap)()
    seg = (C.c_uint32 * cap)()
    if dead:
        rg = (B.HwRange * len(dead))(*[B.HwRange(p, w) for p, w in dead])
        n = rt.lib.acehip_hw_batch_plan_discard(rt.h, arr, len(prog), rg, len(dead), out, launch, seg, cap, SCRATCH)
    else:
        n = rt.lib.acehip_hw_batch_plan(rt.h, arr, len(prog), out, launch, seg, cap, SCRATCH)
    assert 0 <= n <= cap, (n, rt.err())
    return [(out[i].op, out[i].prime_gi, out[i].res, out[i].a, out[i].b, launch[i], seg[i]) for i in range(n)]


class _Lane:
    """what one op of a segment sees: the two most recent results of the segment come from registers (hw_batch_ew_kernel: keyed by
    limb; a result replaces its own limb's entry, otherwise the newer entry becomes the older one), everything else from memory;
    the op's own result is caught instead of stored"""

    def __init__(self, mem, regs):
        self.mem, self.regs, self.out = mem, regs, None

    def __getitem__(self, p):
        for addr, val in self.regs:
            if addr == p:
                return val
        return self.mem[p]

    def __setitem__(self, p, v):
        self.out = v


def _replay(plan, mem, primes, perms, rng):
    by_launch = {}
    for op, gi, res, a, b, la, sg in plan:
        by_launch.setdefault(la, {}).setdefault(sg, []).append((op, gi, res, a, b))
    for la in sorted(by_launch):
        segs = list(by_launch[la].values())
        rng.shuffle(segs)                       # concurrent chains: any order must do
        for ops in segs:
            regs = []                           # [(limb, value)], most recent first, at most two
            for opf, gi, res, a, b in ops:
                op = opf & 0xFF
                for p in (res, a if op != B.HW_ZERO else None, b if op in (B.HW_ADD, B.HW_SUB, B.HW_MUL, B.HW_MULADD) else None):
                    if p is not None and p not in mem:
                        assert p >= SCRATCH, "plan names an address outside the caller's limbs and the scratch arena"
                        mem[p] = ["uninitialised"] * N   # reading it before a write would poison the result
                lane = _Lane(mem, [] if op == B.HW_ROTATE else regs)
                _exec(lane, primes, perms, op, gi, res, a, b)
                # ACEHIP_HW_NOSTORE: the result exists in registers only; memory keeps something a later load must not use
                mem[res] = ["not stored"] * N if opf & B.HW_NOSTORE else lane.out
                if regs and regs[0][0] == res:
                    regs = [(res, lane.out)] + regs[1:]
                else:
                    regs = [(res, lane.out)] + regs[:1]


def _random_program(rt, rng, n_limbs, n_ops, with_rot):
    T = rt.L + rt.K
    prog = []
    perm_keys = [1, 2]
    while len(prog) < n_ops:
        if with_rot and rng.random() < 0.15:
            k = rng.choice(perm_keys)
            for _ in range(rng.randint(1, 6)):
                g = rng.randrange(T)
                r, a = rng.sample(range(g, n_limbs, T), 2)
                prog.append((B.HW_ROTATE, g, BASE + r * SPAN, BASE + a * SPAN, k))
        else:
            for _ in range(rng.randint(1, 30)):
                g = rng.randrange(T)
                r, a, b = (BASE + rng.choice(range(g, n_limbs, T)) * SPAN for _ in range(3))
                op = rng.choice([B.HW_ADD, B.HW_ADD, B.HW_MUL, B.HW_MUL, B.HW_COPY, B.HW_ZERO, B.HW_ZERO, B.HW_SUB,
                                 B.HW_MULADD, B.HW_MULADD, B.HW_MULC, B.HW_ADDC])
                if op in (B.HW_MULC, B.HW_ADDC):
                    b = rng.randrange(rt.primes[g])
                prog.append((op, g, r, a, b))
    return prog[:n_ops]


def _check(rt, prog, n_limbs, seed, dead=()):
    """dead: (address, words) ranges handed to acehip_hw_batch_plan_discard; limbs wholly inside may end up as anything"""
    rng = random.Random(seed)
    T = rt.L + rt.K
    perms = {1: [rng.randrange(N) for _ in range(N)], 2: list(reversed(range(N)))}  # any index table will do
    mem0 = {BASE + i * SPAN: [rng.randrange(rt.primes[i % T]) for _ in range(N)] for i in range(n_limbs)}
    want = {k: list(v) for k, v in mem0.items()}
    for op, gi, res, a, b in prog:
        _exec(want, rt.primes, perms, op, gi, res, a, b)
    plan = _plan(rt, prog, dead)
    given_up = lambda addr: any(p <= addr and addr + SPAN <= p + 8 * w for p, w in dead)  # noqa: E731
    for trial in range(3):                      # three different interleavings of the concurrent segments
        got = {k: list(v) for k, v in mem0.items()}
        _replay(plan, got, rt.primes, perms, random.Random(seed * 7 + trial))
        for addr in mem0:
            if not given_up(addr):
                assert got[addr] == want[addr], (hex(addr), trial)
    return plan


@pytest.mark.parametrize("seed", range(12))
def test_random_programs_replay_to_the_sequential_result(rt, seed):
    rng = random.Random(100 + seed)
    T = rt.L + rt.K
    prog = _random_program(rt, rng, n_limbs=3 * T, n_ops=rng.choice([40, 150, 400]), with_rot=seed % 2 == 0)
    _check(rt, prog, 3 * T, seed)


def test_generated_key_inner_product_is_split_into_independent_chains(rt):
    """resnet20_cifar10_pre.onnx.inc:7011-7036: every product goes through ONE scratch limb; after renaming there must be
    (at least) one chain per limb and component instead of a single serial chain.  Two digits: the first product of every
    accumulator meets the zero fill (0 + a*b = a*b: written straight into the accumulator, the fill dies), the second
    becomes a multiply-add; no product reaches memory except the ones the caller can see in its scratch limb"""
    T = rt.L + rt.K
    at = lambda row, g: BASE + (row * T + g) * SPAN  # noqa: E731
    tmp = at(9, 0)
    prog = [(B.HW_ZERO, 0, at(4, g), 0, 0) for g in range(T)] + [(B.HW_ZERO, 0, at(5, g), 0, 0) for g in range(T)]
    for digit in range(2):
        for g in range(T):
            prog += [(B.HW_MUL, g, tmp, at(2 + 4 * digit, g), at(0, g)), (B.HW_ADD, g, at(4, g), at(4, g), tmp),
                     (B.HW_MUL, g, tmp, at(3 + 4 * digit, g), at(0, g)), (B.HW_ADD, g, at(5, g), at(5, g), tmp)]
    plan = _check(rt, prog, 10 * T, 5)
    kinds = [p[0] & 0xFF for p in plan]
    segs = {}
    for q in plan:
        segs.setdefault((q[5], q[6]), []).append(q)
    # independent chains: one per limb -- the two accumulators of a limb are SIBLINGS (both multiply by the same raised-digit limb) and share
    # a segment, ops in program order, so that the kernel's operand entry loads that limb once (round 5)
    assert len(segs) >= T - 1
    shared_b = sum(1 for ops in segs.values() for x, y in zip(ops, ops[1:]) if x[4] == y[4] and (x[0] & 0xFF) in (B.HW_MUL, B.HW_MULADD)
                   and (y[0] & 0xFF) in (B.HW_MUL, B.HW_MULADD))
    assert shared_b >= 2 * (T - 1)                                  # (per limb: digit 0 and digit 1, the second product right behind the first)
    assert sum(1 for p in plan if p[2] >= SCRATCH) == 0             # no private version is ever written
    assert B.HW_ZERO not in kinds                                   # every fill met its first addend
    # digit 0: 2T products written into their accumulator directly; digit 1: 2T multiply-adds but the very last (product in the
    # scratch limb + add)
    assert kinds.count(B.HW_MUL) == 2 * T + 1 and kinds.count(B.HW_MULADD) == 2 * T - 1
    assert kinds.count(B.HW_COPY) == 0 and kinds.count(B.HW_ADD) == 1
    # only the last product is what the caller finds in its scratch limb afterwards
    in_tmp = [p[0] for p in plan if p[2] == tmp]
    assert in_tmp == [B.HW_MUL]


def test_convolution_taps_keep_the_accumulator_in_registers(rt):
    """The tap loop of a generated convolution (resnet20_cifar10_pre.onnx.inc:1486-1502): per tap and limb  mul tmp.c0, mul tmp.c1,
    add acc.c0, add acc.c1  -- both products BEFORE both accumulations, through one temporary ciphertext.  A product is never
    next to its accumulation in list order, only in its chain: every tap but the last must still become a multiply-add on the
    accumulator (no product stored, the accumulator neither stored nor reloaded between taps)."""
    T = rt.L + rt.K
    at = lambda row, g: BASE + (row * T + g) * SPAN  # noqa: E731
    taps, n_limbs = 5, 3
    # rows: 0/1 acc c0/c1, 2/3 tmp c0/c1, 4+2k / 5+2k the rotated input of tap k (c0 / c1), 20+k the plaintext of tap k
    prog = []
    for k in range(taps):
        for g in range(n_limbs):
            prog += [(B.HW_MUL, g, at(2, g), at(4 + 2 * k, g), at(20 + k, g)), (B.HW_MUL, g, at(3, g), at(5 + 2 * k, g), at(20 + k, g)),
                     (B.HW_ADD, g, at(0, g), at(0, g), at(2, g)), (B.HW_ADD, g, at(1, g), at(1, g), at(3, g))]
    plan = _check(rt, prog, 26 * T, 11)
    kinds = [p[0] & 0xFF for p in plan]
    chains = 2 * n_limbs
    assert kinds.count(B.HW_MULADD) == chains * (taps - 1)           # every tap but the last, for both polynomials of every limb
    assert kinds.count(B.HW_MUL) == chains and kinds.count(B.HW_ADD) == chains   # the last tap: its product stays visible in tmp
    assert sum(1 for p in plan if p[2] >= SCRATCH) == 0              # no private version of tmp is ever written
    for g in range(n_limbs):                                         # one store of each accumulator limb: the final one
        for row in (0, 1):
            stores = [p for p in plan if p[2] == at(row, g) and not (p[0] & B.HW_NOSTORE)]
            assert len(stores) <= 2, stores  # (the multiply-add that hands over to the last tap's plain add, and that add)
    # siblings (round 5): the c0 and the c1 chain of a limb multiply by the same plaintext limb; they share a segment, c0 and c1 of a tap
    # next to each other, so that the kernel's operand entry loads the plaintext limb once per tap instead of twice
    segs = {}
    for q in plan:
        segs.setdefault((q[5], q[6]), []).append(q)
    assert len(segs) == n_limbs
    for ops in segs.values():
        pairs = sum(1 for x, y in zip(ops, ops[1:]) if x[4] == y[4] and x[4] >= at(20, 0))
        assert pairs >= taps - 1, [hex(o[4]) for o in ops]


def test_tails_of_several_rotations_run_in_three_stages(rt):
    """What the rt_ant shim hands over since it keeps ops queued across the direct launches of a key-switch: the TAILS of several
    generated Rotate() calls in one list -- per rotation k and limb:  d0_k += c0 (elementwise), rot_k = gather(d0_k), gather(d1_k),
    then the tap  tmp = rot_k * pt_k, acc += tmp.  In program order every gather cuts the elementwise run, and the accumulator is
    stored and reloaded per rotation.  The stage order (api_hw_batch.cpp hw_stage_order) runs all additions, then all gathers in ONE
    launch, then all taps: the accumulator is stored once.  ACEHIP_HW_STAGES only changes the schedule, never the result (replayed
    against sequential execution by _check)."""
    T = rt.L + rt.K
    at = lambda row, g: BASE + (row * T + g) * SPAN  # noqa: E731
    rots, n_limbs = 4, 3
    # rows: 0/1 acc c0/c1, 2/3 tmp c0/c1, 4 c0 of the source, 10+4k.. d0_k d1_k rot_k.c0 rot_k.c1, 40+k plaintext k
    prog = []
    for k in range(rots):
        d0, d1, r0, r1 = 10 + 4 * k, 11 + 4 * k, 12 + 4 * k, 13 + 4 * k
        for g in range(n_limbs):
            prog.append((B.HW_ADD, g, at(d0, g), at(d0, g), at(4, g)))
        for g in range(n_limbs):
            prog += [(B.HW_ROTATE, g, at(r0, g), at(d0, g), 1 + k % 2), (B.HW_ROTATE, g, at(r1, g), at(d1, g), 1 + k % 2)]
        for g in range(n_limbs):
            prog += [(B.HW_MUL, g, at(2, g), at(r0, g), at(40 + k, g)), (B.HW_MUL, g, at(3, g), at(r1, g), at(40 + k, g)),
                     (B.HW_ADD, g, at(0, g), at(0, g), at(2, g)), (B.HW_ADD, g, at(1, g), at(1, g), at(3, g))]
    plan = _check(rt, prog, 45 * T, 17)
    launches = []
    for p in plan:
        kind = "rot" if (p[0] & 0xFF) == B.HW_ROTATE else "ew"
        if not launches or launches[-1][0] != p[5]:
            launches.append((p[5], kind))
    assert [k for _, k in launches] == ["ew", "rot", "ew"], launches     # (program order: ew rot ew rot ew rot ew rot ew)
    for g in range(n_limbs):
        for row in (0, 1):
            stores = [p for p in plan if p[2] == at(row, g) and not (p[0] & B.HW_NOSTORE)]
            assert len(stores) <= 2, stores                              # not once per rotation


def test_stage_order_respects_every_hazard(rt):
    """gathers and elementwise ops that DO depend on each other keep their order: a gather that reads what an earlier elementwise
    op wrote, an elementwise op that overwrites the source of an earlier gather (write after read), a gather into a limb an earlier
    elementwise op reads, two gathers through the same limb"""
    T = rt.L + rt.K
    at = lambda row, g: BASE + (row * T + g) * SPAN  # noqa: E731
    g = 0
    prog = [(B.HW_ADD, g, at(1, g), at(0, g), at(0, g)),        # x1 = 2 x0
            (B.HW_ROTATE, g, at(2, g), at(1, g), 1),            # x2 = rot(x1)           read after write
            (B.HW_MUL, g, at(1, g), at(0, g), at(0, g)),        # x1 = x0^2              write after read of the gather's source
            (B.HW_ROTATE, g, at(3, g), at(1, g), 2),            # x3 = rot(x1 new)
            (B.HW_ADD, g, at(4, g), at(3, g), at(2, g)),        # x4 = x3 + x2
            (B.HW_ROTATE, g, at(2, g), at(4, g), 1),            # x2 = rot(x4)           gather into a limb an earlier op read
            (B.HW_ROTATE, g, at(5, g), at(2, g), 2),            # x5 = rot(x2)           gather after gather through x2
            (B.HW_SUB, g, at(0, g), at(5, g), at(4, g))]        # x0 = x5 - x4           overwrites the first operand of everything
    _check(rt, prog, 6 * T, 23)


def test_scalar_term_sums_keep_the_accumulator_in_registers(rt):
    """Sums of scalar multiples as the generated activation polynomials and the bootstrap spell them: per term  t = copy(x_k);
    t = t * c_k; acc = acc + t  through one temporary.  The kernel keeps its two most recent results in registers (the temporary
    and the accumulator alternate), so no term may store the temporary and the accumulator reaches memory once."""
    T = rt.L + rt.K
    at = lambda row, g: BASE + (row * T + g) * SPAN  # noqa: E731
    terms = 7
    prog = []
    for g in range(2):
        for k in range(terms):
            prog += [(B.HW_COPY, g, at(1, g), at(4 + k, g), 0), (B.HW_MULC, g, at(1, g), at(1, g), 3 + k),
                     (B.HW_ADD, g, at(0, g), at(0, g), at(1, g))]
    plan = _check(rt, prog, 12 * T, 13)
    for g in range(2):
        acc_stores = [p for p in plan if p[2] == at(0, g) and not (p[0] & B.HW_NOSTORE)]
        assert len(acc_stores) == 1, acc_stores
        # the temporary is visible to the caller afterwards: its last version is stored, nothing else
        tmp_stores = [p for p in plan if (p[2] == at(1, g) or p[2] >= SCRATCH) and not (p[0] & B.HW_NOSTORE)]
        assert len(tmp_stores) <= 2, tmp_stores


def test_dead_zero_fills_and_copies_are_dropped(rt):
    T = rt.L + rt.K
    at = lambda row, g: BASE + (row * T + g) * SPAN  # noqa: E731
    prog = []
    for g in range(T):   # Alloc_poly zero fill, then the Hw_* loop overwrites the limb: the fill is dead
        prog += [(B.HW_ZERO, 0, at(2, g), 0, 0), (B.HW_COPY, 0, at(3, g), at(0, g), 0), (B.HW_MUL, g, at(2, g), at(0, g), at(1, g)),
                 (B.HW_ADD, g, at(3, g), at(0, g), at(1, g))]
    plan = _check(rt, prog, 4 * T, 6)
    assert not any(p[0] in (B.HW_ZERO, B.HW_COPY) for p in plan)
    assert len(plan) == 2 * T


@pytest.mark.parametrize("seed", range(16))
def test_random_programs_with_given_up_memory(rt, seed):
    """acehip_hw_batch_discard: every limb outside the given-up ranges ends up as after the plain list, whatever the library
    skips or keeps in registers; ranges are limb-aligned blocks, ragged ones (half a limb at either end) and single limbs"""
    rng = random.Random(500 + seed)
    T = rt.L + rt.K
    n_limbs = 3 * T
    prog = _random_program(rt, rng, n_limbs=n_limbs, n_ops=rng.choice([40, 150, 400]), with_rot=seed % 2 == 0)
    dead, i = [], 0
    while i < n_limbs:
        if rng.random() < 0.35:
            n = rng.randint(1, 4)
            lo = BASE + i * SPAN - (SPAN // 2 if rng.random() < 0.3 and i > 0 and not dead_ends_at(dead, BASE + i * SPAN) else 0)
            hi = BASE + min(n_limbs, i + n) * SPAN + (SPAN // 2 if rng.random() < 0.3 else 0)
            dead.append((lo, (hi - lo) // 8))
            i += n + 1
        else:
            i += 1
    plan = _check(rt, prog, n_limbs, seed, dead)
    plain = _plan(rt, prog)
    assert len(plan) <= len(plain)


@pytest.mark.parametrize("seed", range(12))
def test_zero_heavy_programs(rt, seed):
    """a quarter of the ops are zero fills, few limbs: ops on just-cleared operands are rewritten (0 + x -> copy, 0 * x -> fill,
    0 + a*b -> product) and the rewritten list must still replay to the sequential result, with and without given-up limbs"""
    rng = random.Random(77000 + seed)
    T = rt.L + rt.K
    n_limbs = rng.choice([2, 3]) * T
    prog = []
    for _ in range(rng.choice([30, 120, 500])):
        g = rng.randrange(T)
        r, a, b = (BASE + rng.choice(range(g, n_limbs, T)) * SPAN for _ in range(3))
        op = rng.choice([B.HW_ADD, B.HW_ADD, B.HW_MUL, B.HW_COPY, B.HW_ZERO, B.HW_ZERO, B.HW_ZERO, B.HW_SUB, B.HW_MULADD,
                         B.HW_MULADD, B.HW_MULC, B.HW_ADDC])
        if op in (B.HW_MULC, B.HW_ADDC):
            b = rng.randrange(rt.primes[g])
        prog.append((op, g, r, a, b))
    _check(rt, prog, n_limbs, seed)
    _check(rt, prog, n_limbs, seed, [(BASE + i * SPAN, N) for i in range(n_limbs) if rng.random() < 0.3])


def dead_ends_at(dead, addr):
    return any(p + 8 * w > addr - SPAN for p, w in dead)


def test_temporaries_of_freed_blocks_stay_in_registers(rt):
    """the tensor product of the generated code: t = a*b; acc = acc + t with t in a block the program has freed by the time
    the list is handed over: the product is consumed from registers and never stored; an op that feeds only freed memory
    is not run at all; a freed limb that a later rotation run reads is still produced"""
    T = rt.L + rt.K
    at = lambda row, g: BASE + (row * T + g) * SPAN  # noqa: E731
    prog = []
    for g in range(T):
        prog += [(B.HW_MUL, g, at(3, g), at(0, g), at(1, g)),       # t (row 3: freed)
                 (B.HW_ADD, g, at(2, g), at(2, g), at(3, g)),       # acc += t
                 (B.HW_MULC, g, at(4, g), at(0, g), 5),             # row 4: freed, read by nobody
                 (B.HW_SUB, g, at(5, g), at(0, g), at(1, g))]       # row 5: freed, read by the rotation below
    prog += [(B.HW_ROTATE, g, at(6, g), at(5, g), 1) for g in range(T)]
    dead = [(at(3, 0), 3 * T * N)]                                  # rows 3, 4, 5
    plan = _check(rt, prog, 7 * T, 11, dead)
    kinds = [p[0] & 0xFF for p in plan]
    assert B.HW_MULC not in kinds                                   # fed only freed memory
    assert kinds.count(B.HW_SUB) == T and kinds.count(B.HW_ROTATE) == T
    for p in plan:
        if p[0] & 0xFF == B.HW_SUB:
            assert not p[0] & B.HW_NOSTORE                          # the rotation loads it from memory
        if p[0] & 0xFF == B.HW_MUL:
            assert p[0] & B.HW_NOSTORE                              # straight into the add
    # without the hint everything is computed and stored
    plain = _plan(rt, prog)
    assert len(plain) == len(prog) and not any(p[0] & B.HW_NOSTORE for p in plain)


def test_discard_argument_errors(rt):
    prog = (B.HwOp * 1)(B.HwOp(B.HW_ADD, 0, BASE, BASE, BASE))
    out, la, sg = (B.HwOp * 4)(), (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
    rg = (B.HwRange * 2)(B.HwRange(BASE, 2 * N), B.HwRange(BASE + SPAN, N))      # overlapping
    assert rt.lib.acehip_hw_batch_plan_discard(rt.h, prog, 1, rg, 2, out, la, sg, 4, SCRATCH) < 0
    assert rt.lib.acehip_hw_batch_plan_discard(rt.h, prog, 1, None, 2, out, la, sg, 4, SCRATCH) < 0
    assert rt.lib.acehip_hw_batch_plan_discard(rt.h, prog, 1, None, 0, out, la, sg, 4, SCRATCH) == 1


def test_partially_overlapping_limbs_run_one_by_one(rt):
    prog = [(B.HW_ADD, 0, BASE, BASE, BASE + SPAN // 2), (B.HW_MUL, 0, BASE + 2 * SPAN, BASE + SPAN // 2, BASE)]
    plan = _plan(rt, prog)
    # no reordering: one launch per op (plus the private copies of operands that overlap a result)
    assert [p[5] for p in plan] == sorted(p[5] for p in plan) and len({p[5] for p in plan}) == len(plan)


def test_argument_errors(rt):
    bad = (B.HwOp * 1)(B.HwOp(B.HW_ADD, 99, BASE, BASE, BASE))
    out, la, sg = (B.HwOp * 4)(), (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
    assert rt.lib.acehip_hw_batch_plan(rt.h, bad, 1, out, la, sg, 4, SCRATCH) < 0
    assert rt.lib.acehip_hw_batch_plan(rt.h, bad, 1, out, la, sg, 4, 0) < 0
