"""Multi-GPU glue for bench.py: one process per GPU, independent image streams (replicas).

The path shards by unit (image / ciphertext): every rank owns its own context, keys and inputs, there is no
data-path collective (SURVEY 8e "Replicas"; the reference's own parallel axis is images,
rtlib/ant/dataset/resnet_cifar.main.inc:77-116).  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on
CPU for the tests) only carries the barrier and the max-reduce of the timed region.
"""
import os


class Ranks:
    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.backend = backend
        if self.world > 1:
            import torch
            import torch.distributed as dist

            self.backend = backend or "nccl"
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        """max of a python float over all ranks (the timed region is as long as the slowest rank)."""
        if self.dist is None:
            return float(value)
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch

        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def shard(self, n_units):
        """contiguous slice of `n_units` independent units owned by this rank (strong-scaling helper)."""
        per, rem = divmod(n_units, self.world)
        lo = self.rank * per + min(self.rank, rem)
        return range(lo, lo + per + (1 if self.rank < rem else 0))

    def aggregate_throughput(self, units_per_rank, elapsed_s):
        """whole-job throughput: all ranks' units / slowest rank's time."""
        total = self.sum_over_ranks(units_per_rank)
        return total / self.max_over_ranks(elapsed_s)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
