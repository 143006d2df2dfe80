"""Limb-sharded execution (BASELINE configs[4]): harness around the C-ABI phases of include/acehip.h
(acehip_shard_*).  The arithmetic and every launch live in the library (csrc/api_shard.cpp,
csrc/shard.hip); this module only moves the exchange buffers:

  * `TorchComm`: one process per GPU, torch.distributed all_gather_into_tensor / broadcast on device tensors
    (backend "nccl" = RCCL over xGMI on a node; "gloo" on CPU tensors in the layout test),
  * `LocalComm`: `world` ranks simulated in ONE process on one GPU (tests, single-GPU bench): the gather is a set of
    device-to-device copies.

Partition: limb gi (q_i: gi = i, p_j: gi = L + j) lives on rank gi % world, packed in ascending gi, for polynomials,
switch keys and tables alike.  Exchanges per key-switch: all-gather of the coefficient-domain input (pad_q limbs per
rank) and of the coefficient-domain p-limbs of both accumulators (2*pad_p per rank); per rescale: broadcast of two limbs.
"""
import numpy as np


def owner(gi, world):
    return gi % world


def owned_q(L, world, rank, level):
    return [i for i in range(min(level, L)) if owner(i, world) == rank]


def owned_p(L, K, world, rank):
    return [j for j in range(K) if owner(L + j, world) == rank]


def pack_key(key_full, L, K, world, rank):
    """[nd][2][L+K][N] -> this rank's [nd][2][n_own][N] (owned q-limbs of the full chain, then owned p-limbs)"""
    idx = owned_q(L, world, rank, L) + [L + j for j in owned_p(L, K, world, rank)]
    return np.ascontiguousarray(key_full[:, :, idx, :])


class Shard:
    """one rank's handle (acehip_shard) plus its exchange buffers, allocated once"""

    def __init__(self, rt, rank, world):
        self.rt, self.rank, self.world = rt, rank, world
        self.h = rt.lib.acehip_shard_create(rt.h, rank, world)
        if not self.h:
            raise RuntimeError(rt.err())
        self.N, self.L, self.K = rt.N, rt.L, rt.K
        self.pad_q_max = rt.lib.acehip_shard_pad_q(self.h, self.L)
        self.pad_p = rt.lib.acehip_shard_pad_p(self.h)
        self.send1 = rt.buf(max(self.pad_q_max, 1) * self.N)
        self.send2 = rt.buf(max(2 * self.pad_p, 1) * self.N)
        self.last = rt.buf(2 * self.N)

    def num_q(self, level):
        return self.rt.lib.acehip_shard_num_q(self.h, level)

    def pad_q(self, level):
        return self.rt.lib.acehip_shard_pad_q(self.h, level)

    def close(self):
        for b in (self.send1, self.send2, self.last):
            b.free()
        self.rt.lib.acehip_shard_destroy(self.h)

    # the three phases of a key-switch and the two halves of a rescale; gathered buffers are device addresses
    def ks_phase1(self, x_own_ptr, level):
        self.rt.check(self.rt.lib.acehip_shard_ks_phase1(self.h, self.send1.ptr, x_own_ptr, level, None))

    def ks_phase2(self, gathered_ptr, x_own_ptr, key_own_ptr, level):
        self.rt.check(self.rt.lib.acehip_shard_ks_phase2(self.h, self.send2.ptr, gathered_ptr, x_own_ptr, key_own_ptr, level, None))

    def ks_phase3(self, out0_ptr, out1_ptr, gathered2_ptr, level):
        self.rt.check(self.rt.lib.acehip_shard_ks_phase3(self.h, out0_ptr, out1_ptr, gathered2_ptr, level, None))

    def rescale_send(self, c0_ptr, c1_ptr, level):
        rc = self.rt.lib.acehip_shard_rescale_send(self.h, self.last.ptr, c0_ptr, c1_ptr, level, None)
        if rc < 0:
            self.rt.check(rc)
        return rc == 1

    def rescale_apply(self, out0_ptr, out1_ptr, c0_ptr, c1_ptr, last_ptr, level):
        self.rt.check(self.rt.lib.acehip_shard_rescale_apply(self.h, out0_ptr, out1_ptr, c0_ptr, c1_ptr, last_ptr, level, None))


class LocalWorld:
    """`world` ranks of one process on one GPU: every exchange is device-to-device copies into a gathered buffer.  This is
    the arithmetic of a node run with the collectives replaced by copies (what tests/test_gpu_shard.py checks bit for bit
    against the unsharded oracle, and what `bench.py --mode shard` times on a single GPU)."""

    def __init__(self, rt, world):
        self.rt, self.world = rt, world
        self.shards = [Shard(rt, r, world) for r in range(world)]
        s0 = self.shards[0]
        self.N, self.L, self.K = rt.N, rt.L, rt.K
        self.g1 = rt.buf(max(world * s0.pad_q_max, 1) * self.N)
        self.g2 = rt.buf(max(world * 2 * s0.pad_p, 1) * self.N)

    def close(self):
        for s in self.shards:
            s.close()
        self.g1.free()
        self.g2.free()

    def split(self, x_full, level):
        """[level][N] numpy -> per rank DeviceBuffer with its owned limbs (at least one limb of storage)"""
        out = []
        for r in range(self.world):
            idx = owned_q(self.L, self.world, r, level)
            out.append(self.rt.to_device(x_full[idx] if idx else np.zeros((1, self.N), dtype=np.uint64)))
        return out

    def join(self, bufs, level):
        full = np.zeros((level, self.N), dtype=np.uint64)
        for r, b in enumerate(bufs):
            idx = owned_q(self.L, self.world, r, level)
            if idx:
                full[idx] = b.download()[: len(idx) * self.N].reshape(len(idx), self.N)
        return full

    def key_switch(self, x_own, key_own, out0, out1, level):
        """x_own / key_own / out0 / out1: per-rank DeviceBuffers (packed owned limbs); all launches on this thread's stream"""
        rt, N, nb = self.rt, self.N, self.N * 8
        W = self.world
        pad_q, pad_p = self.shards[0].pad_q(level), self.shards[0].pad_p
        for r, s in enumerate(self.shards):
            s.ks_phase1(x_own[r].ptr, level)
            n = s.num_q(level)
            if n:
                rt.check(rt.lib.acehip_memcpy_d2d(self.g1.at(r * pad_q * N), s.send1.ptr, n * nb, None))
        for r, s in enumerate(self.shards):
            s.ks_phase2(self.g1.ptr, x_own[r].ptr, key_own[r].ptr, level)
            if pad_p:
                rt.check(rt.lib.acehip_memcpy_d2d(self.g2.at(r * 2 * pad_p * N), s.send2.ptr, 2 * pad_p * nb, None))
        for r, s in enumerate(self.shards):
            s.ks_phase3(out0[r].ptr, out1[r].ptr, self.g2.ptr, level)

    def rescale(self, c0_own, c1_own, out0, out1, level):
        sender = None
        for r, s in enumerate(self.shards):
            if s.rescale_send(c0_own[r].ptr, c1_own[r].ptr, level):
                sender = s
        assert sender is not None and sender.rank == owner(level - 1, self.world)
        for r, s in enumerate(self.shards):
            s.rescale_apply(out0[r].ptr, out1[r].ptr, c0_own[r].ptr, c1_own[r].ptr, sender.last.ptr, level)


class TorchComm:
    """torch.distributed communicator (one process per GPU): exchange buffers are torch tensors whose device addresses go to
    the C ABI; "nccl" is RCCL over xGMI, "gloo" works on CPU tensors (layout test)."""

    def __init__(self, dist, device):
        import torch

        self.dist, self.torch, self.device = dist, torch, device
        self.world = dist.get_world_size()

    def all_gather(self, local):
        out = self.torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(out, local.contiguous())
        return out

    def broadcast(self, t, src):
        self.dist.broadcast(t, src=src)
        return t


class RankRunner:
    """one rank of a node run: the C-ABI phases of this rank's shard, torch tensors as exchange buffers (allocated once),
    RCCL collectives between the phases.  Everything is issued on torch's current stream: the library is handed that
    stream, so launches and collectives are ordered without host synchronisation."""

    def __init__(self, rt, comm, rank):
        import torch

        self.rt, self.comm, self.rank, self.torch = rt, comm, rank, torch
        self.sh = Shard(rt, rank, comm.world)
        N, dev = rt.N, comm.device
        self.t_send1 = torch.zeros((max(self.sh.pad_q_max, 1), N), dtype=torch.int64, device=dev)
        self.t_send2 = torch.zeros((max(2 * self.sh.pad_p, 1), N), dtype=torch.int64, device=dev)
        self.t_last = torch.zeros((2, N), dtype=torch.int64, device=dev)

    def _stream(self):
        """The stream launches and collectives share.  It must be an EXPLICIT stream: libacehip is built with
        -fgpu-default-stream=per-thread, so handle 0 would mean "the calling thread's own stream" inside the library while torch
        (and RCCL) mean the legacy stream by it -- two different queues ordered only by implicit legacy-stream synchronisation.
        The runner therefore works under a stream of its own (see run()) and refuses a null handle."""
        if self.comm.device == "cpu":
            return None
        h = self.torch.cuda.current_stream().cuda_stream
        if h == 0:
            raise RuntimeError("RankRunner needs an explicit torch.cuda.Stream (use `with runner.stream():`): the default stream "
                               "is not the stream the library launches on")
        return h

    def stream(self):
        """context manager: the runner's own stream becomes torch's current stream (collectives and C-ABI launches go there)"""
        if getattr(self, "_own_stream", None) is None:
            self._own_stream = self.torch.cuda.Stream()
        return self.torch.cuda.stream(self._own_stream)

    def key_switch(self, x_own_ptr, key_own_ptr, out0_ptr, out1_ptr, level):
        rt, lib, st = self.rt, self.rt.lib, self._stream()
        pad_q = self.sh.pad_q(level)
        rt.check(lib.acehip_shard_ks_phase1(self.sh.h, self.t_send1.data_ptr(), x_own_ptr, level, st))
        g1 = self.comm.all_gather(self.t_send1[:max(pad_q, 1)])
        rt.check(lib.acehip_shard_ks_phase2(self.sh.h, self.t_send2.data_ptr(), g1.data_ptr(), x_own_ptr, key_own_ptr, level, st))
        g2 = self.comm.all_gather(self.t_send2)
        rt.check(lib.acehip_shard_ks_phase3(self.sh.h, out0_ptr, out1_ptr, g2.data_ptr(), level, st))
        return g1, g2  # kept alive by the caller until the stream has consumed them

    def rescale(self, c0_ptr, c1_ptr, out0_ptr, out1_ptr, level):
        rt, lib, st = self.rt, self.rt.lib, self._stream()
        rc = lib.acehip_shard_rescale_send(self.sh.h, self.t_last.data_ptr(), c0_ptr, c1_ptr, level, st)
        if rc < 0:
            rt.check(rc)
        self.comm.broadcast(self.t_last, owner(level - 1, self.comm.world))
        rt.check(lib.acehip_shard_rescale_apply(self.sh.h, out0_ptr, out1_ptr, c0_ptr, c1_ptr, self.t_last.data_ptr(), level, st))

    def close(self):
        self.sh.close()
