"""Limb-sharded hybrid key-switch (BASELINE configs[4]): the RNS limbs of one ciphertext are spread over the
GPUs of a node and only the base conversions exchange data.

Partition: the limb with global prime index gi (q_i: gi = i, p_j: gi = L + j) lives on rank gi % G, for polynomials, switch
keys and twiddle tables alike (ownership does not depend on the level).  One key-switch at level l on rank r:

  1. iNTT of the owned q-limbs of the input                                   (local)
  2. ALL-GATHER of those coefficient-domain limbs: every rank gets all l      (exchange 1: l x 512 KiB at N = 2^16)
  3. per digit d: base conversion of digit d onto the OWNED complement limbs (acehip_base_conv), NTT; the owned limbs
     of the digit itself pass through                                         (local)
  4. key inner product over the owned limbs: acc_c = sum_d key_c[d] * ext_d    (local, acehip_hw_batch)
  5. iNTT of the owned p-limbs of both accumulators                           (local)
  6. ALL-GATHER of the 2K coefficient-domain p-limbs                          (exchange 2: 2K x 512 KiB)
  7. base conversion P -> owned q-limbs, NTT, out = (acc - conv) * P^-1        (local)

The arithmetic is the same as acehip_key_switch (Decompose_modup polynomial.c:1241-1335, Multiply_add :148-183,
Reduce_rns_base :928-967): gathering every rank's output limbs reproduces it bit for bit (tests/test_gpu_shard.py runs
G simulated ranks on one GPU).  The exchange is a pluggable communicator: `TorchComm` = torch.distributed all_gather
(backend "nccl" = RCCL over xGMI on a node, "gloo" in the CPU test of the gather layout), `LocalComm` = G ranks of one
process copying device to device.  At 1.4 s per image and ~5000 key-switches per image a sharded key-switch has
~100 us per exchange to amortise, so this path buys latency for one image, never throughput.
"""
import numpy as np

from . import binding as B


def owner(gi, world):
    return gi % world


class LimbShard:
    """who owns what at a given level: q-limbs are named by i < level, p-limbs by L + j"""

    def __init__(self, L, K, world, rank):
        self.L, self.K, self.world, self.rank = L, K, world, rank

    def q_owned(self, level, rank=None):
        r = self.rank if rank is None else rank
        return [i for i in range(level) if owner(i, self.world) == r]

    def p_owned(self, rank=None):
        r = self.rank if rank is None else rank
        return [j for j in range(self.K) if owner(self.L + j, self.world) == r]

    def max_q(self, level):
        return max(len(self.q_owned(level, r)) for r in range(self.world))

    def max_p(self):
        return max(len(self.p_owned(r)) for r in range(self.world))


def gather_slots(per_rank_items, pad):
    """rank-major layout of an all_gather with `pad` slots per rank -> {item: slot}"""
    return {item: r * pad + k for r, items in enumerate(per_rank_items) for k, item in enumerate(items)}


class ShardedKeySwitch:
    """one rank of the sharded key-switch, written as a generator: it yields ("gather", device_ptr, n_limbs_valid, pad)
    at the two exchange points and is sent the device address of the rank-major gathered buffer [world*pad][N]."""

    def __init__(self, rt, rank, world):
        self.rt, self.rank, self.world = rt, rank, world
        self.sh = LimbShard(rt.L, rt.K, world, rank)
        self.N = rt.N
        self.pinv = rt.table(23)  # P^-1 mod q_i

    # ---- helpers on single limbs at arbitrary addresses ----
    def _ntt(self, ptr, gi, inverse):
        rt, nb = self.rt, self.N * 8
        fn = rt.lib.acehip_ntt_inverse if inverse else rt.lib.acehip_ntt_forward
        if gi < rt.L:
            rt.check(fn(rt.h, ptr - gi * nb, rt.L, gi, 1, None))
        else:
            rt.check(fn(rt.h, ptr - (gi - rt.L) * nb, 0, gi - rt.L, 1, None))

    def run(self, level, x_own, key_limb):
        """x_own: DeviceBuffer with the owned q-limbs of the input (ascending i), NTT domain.
        key_limb(d, comp, gi) -> device address of that key limb (owned gi only).
        Returns (out0, out1): DeviceBuffers with the owned q-limbs of the two outputs."""
        rt, sh, N, nb = self.rt, self.sh, self.N, self.N * 8
        L, K, G = rt.L, rt.K, self.world
        q_own, p_own = sh.q_owned(level), sh.p_owned()
        nd = rt.num_decomp(level)
        alpha = rt.alpha
        # 1. owned q-limbs to the coefficient domain
        pad_q = sh.max_q(level)
        coef = rt.buf(max(pad_q, 1) * N)
        if q_own:
            rt.check(rt.lib.acehip_memcpy_d2d(coef.ptr, x_own.ptr, len(q_own) * nb, None))
        for k, i in enumerate(q_own):
            self._ntt(coef.at(k * N), i, True)
        # 2. exchange 1
        gathered = yield ("gather", coef.ptr, len(q_own), pad_q)
        slot = gather_slots([sh.q_owned(level, r) for r in range(G)], pad_q)
        full = rt.buf(level * N)  # position order
        rt.hw_batch([(B.HW_COPY, 0, full.at(i * N), gathered + slot[i] * nb, None) for i in range(level)])
        # 3. per digit: conversion onto the owned complement limbs, NTT; own digit limbs pass through
        own_pos = [(i, i) for i in q_own] + [(level + j, L + j) for j in p_own]  # (position at this level, global prime)
        ext = {}      # (d, position) -> device address
        ext_bufs = []
        for d in range(nd):
            start, n2 = alpha * d, min(alpha, level - alpha * d)
            tgt = [(p, gi) for p, gi in own_pos if not (start <= p < start + n2)]
            if tgt:
                buf = rt.buf(len(tgt) * N)
                ext_bufs.append(buf)
                pos = np.asarray([p for p, _ in tgt], dtype=np.uint32)
                rt.check(rt.lib.acehip_base_conv(rt.h, buf.ptr, full.at(start * N), level, d, pos.ctypes.data, len(tgt), None))
                for k, (p, gi) in enumerate(tgt):
                    self._ntt(buf.at(k * N), gi, False)
                    ext[(d, p)] = buf.at(k * N)
            for k, i in enumerate(q_own):
                if start <= i < start + n2:
                    ext[(d, i)] = x_own.at(k * N)
        # 4. key inner product over the owned limbs
        acc = [rt.buf(max(len(own_pos), 1) * N) for _ in range(2)]
        ops = []
        for k, (p, gi) in enumerate(own_pos):
            for comp in range(2):
                for d in range(nd):
                    ops.append((B.HW_MUL if d == 0 else B.HW_MULADD, gi, acc[comp].at(k * N), key_limb(d, comp, gi), ext[(d, p)]))
        if ops:
            rt.hw_batch(ops)
        # 5. owned p-limbs of both accumulators to the coefficient domain
        pad_p = sh.max_p()
        pco = rt.buf(max(2 * pad_p, 1) * N)  # [comp][pad_p]
        for comp in range(2):
            for k, j in enumerate(p_own):
                src = acc[comp].at((len(q_own) + k) * N)
                rt.check(rt.lib.acehip_memcpy_d2d(pco.at((comp * pad_p + k) * N), src, nb, None))
                self._ntt(pco.at((comp * pad_p + k) * N), L + j, True)
        # 6. exchange 2 (both accumulators in one gather: 2 * pad_p slots per rank)
        gathered = yield ("gather", pco.ptr, 2 * pad_p, 2 * pad_p)
        pslot = gather_slots([[(comp, j) for comp in range(2) for j in sh.p_owned(r) + [None] * (pad_p - len(sh.p_owned(r)))]
                              for r in range(G)], 2 * pad_p)
        pfull = rt.buf(2 * K * N)
        rt.hw_batch([(B.HW_COPY, 0, pfull.at((comp * K + j) * N), gathered + pslot[(comp, j)] * nb, None)
                     for comp in range(2) for j in range(K)])
        # 7. conversion P -> owned q-limbs, NTT, tail
        outs = [rt.buf(max(len(q_own), 1) * N) for _ in range(2)]
        if q_own:
            pos = np.asarray(q_own, dtype=np.uint32)
            ops = []
            for comp in range(2):
                rt.check(rt.lib.acehip_base_conv(rt.h, outs[comp].ptr, pfull.at(comp * K * N), level, -1, pos.ctypes.data, len(q_own), None))
                for k, i in enumerate(q_own):
                    self._ntt(outs[comp].at(k * N), i, False)
                    o = outs[comp].at(k * N)
                    ops += [(B.HW_SUB, i, o, acc[comp].at(k * N), o), (B.HW_MULC, i, o, o, int(self.pinv[i]))]
            rt.hw_batch(ops)
        rt.sync()
        for b in [coef, full, pco, pfull] + ext_bufs + acc:
            b.free()
        return outs[0], outs[1]


class LocalComm:
    """G ranks of ONE process on one GPU (tests): the gather is a set of device-to-device copies."""

    def __init__(self, rt):
        self.rt = rt

    def all_gather(self, requests):
        """requests: per rank (ptr, n_valid, pad) -> per rank device address of the rank-major gathered buffer"""
        rt, nb = self.rt, self.rt.N * 8
        pad = requests[0][2]
        out = rt.buf(max(len(requests) * pad, 1) * rt.N)
        for r, (ptr, n_valid, _) in enumerate(requests):
            if n_valid:
                rt.check(rt.lib.acehip_memcpy_d2d(out.at(r * pad * rt.N), ptr, n_valid * nb, None))
        rt.sync()
        return out


def run_local(rt, world, level, x_full, key_full):
    """simulate `world` ranks on one GPU: x_full [level][N] and key_full [nd][2][L+K][N] numpy arrays -> (out0, out1)
    [level][N] assembled from the ranks' owned output limbs"""
    N, L, K = rt.N, rt.L, rt.K
    comm = LocalComm(rt)
    d_key = rt.to_device(key_full)
    T = L + K
    gens, xs = [], []
    for r in range(world):
        ks = ShardedKeySwitch(rt, r, world)
        q_own = ks.sh.q_owned(level)
        x_own = rt.to_device(x_full[q_own] if q_own else np.zeros((1, N), dtype=np.uint64))
        xs.append(x_own)
        gens.append(ks.run(level, x_own, lambda d, comp, gi: d_key.at(((d * 2 + comp) * T + gi) * N)))
    reqs = [next(g) for g in gens]
    results = [None] * world
    held = []
    while any(r is not None for r in reqs):
        gathered = comm.all_gather([(q[1], q[2], q[3]) for q in reqs])
        held.append(gathered)
        nxt = []
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send(gathered.ptr))
            except StopIteration as stop:
                results[r] = stop.value
                nxt.append(None)
        reqs = nxt
        if all(q is None for q in reqs):
            break
    out0 = np.zeros((level, N), dtype=np.uint64)
    out1 = np.zeros((level, N), dtype=np.uint64)
    for r in range(world):
        q_own = LimbShard(L, K, world, r).q_owned(level)
        o0, o1 = results[r]
        if q_own:
            out0[q_own] = o0.download((max(len(q_own), 1), N))[:len(q_own)]
            out1[q_own] = o1.download((max(len(q_own), 1), N))[:len(q_own)]
        o0.free()
        o1.free()
    for b in held + xs + [d_key]:
        b.free()
    return out0, out1


class TorchComm:
    """torch.distributed communicator (one process per GPU): the exchange buffers are torch tensors whose device
    addresses are handed to the C ABI; backend "nccl" is RCCL over xGMI, "gloo" works on CPU tensors (layout test)."""

    def __init__(self, dist, device):
        import torch

        self.dist, self.torch, self.device = dist, torch, device
        self.world = dist.get_world_size()

    def all_gather_tensor(self, local):
        """local: [pad, N] int64 tensor -> [world*pad, N] rank-major"""
        out = self.torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(out, local.contiguous())
        return out


def run_rank(ks, gen, comm):
    """drive one rank's generator with a TorchComm: each exchange copies the rank's limbs into a torch tensor, all-gathers
    it (RCCL on a node) and hands the gathered tensor's device address back.  Returns the generator's result."""
    rt, N, torch = ks.rt, ks.N, comm.torch
    keep = []
    try:
        req = next(gen)
        while True:
            _, ptr, n_valid, pad = req
            local = torch.zeros((max(pad, 1), N), dtype=torch.int64, device=comm.device)
            if n_valid:
                rt.check(rt.lib.acehip_memcpy_d2d(local.data_ptr(), ptr, n_valid * N * 8, None))
            rt.sync()
            torch.cuda.synchronize()
            gathered = comm.all_gather_tensor(local)
            torch.cuda.synchronize()
            keep += [local, gathered]
            req = gen.send(gathered.data_ptr())
    except StopIteration as stop:
        return stop.value
