"""Shard-by-frame helper for multi-GPU batches (one process per GPU).

Every hot-path function is a pure function of one image, so a batch shards by
frame with NO data-path collective: frame f lives on, is generated on and is processed by exactly
one GPU and never moves.  torch.distributed (backend "nccl" == RCCL over xGMI on ROCm, "gloo" in
the CPU tests) is used only for KB-scale control traffic: the timing barrier, the max-over-ranks
of the elapsed time and gathering per-frame results (thresholds / counts / checksums).
"""
import os


def frame_range(rank, world, total):
    """contiguous, balanced block of frame indices owned by `rank`: [lo, hi)"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def owner_of(frame, world, total):
    base, rem = divmod(total, world)
    cut = rem * (base + 1)
    return frame // (base + 1) if frame < cut else rem + (frame - cut) // max(base, 1)


class Sharder:
    """rank/world bookkeeping + the three control-plane collectives the batch driver needs"""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend=backend or "nccl", rank=self.rank,
                                        world_size=self.world)
            self.dist = dist
        self.backend = backend or ("nccl" if self.world > 1 else None)

    def _dev(self):
        import torch
        return torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def ranks_seen(self):
        """all-reduce of 1 over the job's backend (RCCL on the GPUs): the number of ranks that take part"""
        return int(round(self.sum_over_ranks(1.0)))

    def all_gather_frames(self, local, total):
        """local: 1-D tensor with this rank's per-frame results (frame_range order, equal-length
        shards padded); returns the `total` per-frame values in global frame order"""
        import torch
        if not self.dist:
            return local[:total].clone()
        per = (total + self.world - 1) // self.world
        pad = torch.zeros(per, dtype=local.dtype, device=self._dev())  # gloo gathers on the host
        pad[:local.numel()] = local.to(pad.device)
        out = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(out, pad)
        parts = []
        for r in range(self.world):
            lo, hi = frame_range(r, self.world, total)
            parts.append(out[r][:hi - lo])
        return torch.cat(parts).to(local.device)

    def close(self):
        if self.dist and self.dist.is_initialized():
            self.dist.destroy_process_group()
