"""Shard-by-frame helper for multi-GPU batches (one process per GPU).

Every hot-path function is a pure function of one image, so a batch shards by
frame with NO data-path collective: frame f lives on, is generated on and is processed by exactly
one GPU and never moves.  torch.distributed (backend "nccl" == RCCL over xGMI on ROCm, "gloo" in
the CPU tests) is used only for control traffic and results (SURVEY.md 8e's four collectives): (1) the
broadcast of the flattened cascade blob (~7 KB) from rank 0, (2) the all-gather of per-frame results
(thresholds / counts / checksums), (3) the variable-length gather of the gs_rect / gs_keypoint lists,
(4) the timing barrier and the max-over-ranks of the elapsed time.  Pixel planes never cross GPUs.

GS_BENCH_FORCE_DIST=1 (or force=True) initialises the process group at world 1 too, so that the very
same RCCL calls can be exercised on a one-GPU box.
"""
import os
import socket


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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

    def __init__(self, backend=None, force=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.dist = None
        if force is None:
            force = os.environ.get("GS_BENCH_FORCE_DIST", "") not in ("", "0")
        if self.world > 1 or force:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                if "MASTER_PORT" not in os.environ:  # only possible at world 1 (no launcher): any free port
                    os.environ["MASTER_PORT"] = str(_free_port())
                dist.init_process_group(backend=backend or "nccl", rank=self.rank,
                                        world_size=self.world)
            self.dist = dist
        self.backend = backend or ("nccl" if self.dist else None)

    def _dev(self):
        import torch
        return torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.dist:
            if self.backend == "nccl":  # name the rank's GPU: without it RCCL guesses the device from the current one
                self.dist.barrier(device_ids=[self.local_rank])
            else:
                self.dist.barrier()

    def max_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(self, value):
        if not self.dist:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
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

    def broadcast_bytes(self, data, root=0):
        """SURVEY 8(e) collective (1): rank `root` hands every rank the same byte string (the flattened
        cascade blob); `data` is only read on `root`.  Two broadcasts: the length, then the bytes."""
        if not self.dist:
            return bytes(data)
        import torch
        n = torch.tensor([len(data) if self.rank == root else 0], dtype=torch.int64, device=self._dev())
        self.dist.broadcast(n, src=root)
        if int(n.item()) == 0:  # nothing to send (torch.frombuffer refuses an empty buffer)
            return b""
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=self._dev())
        if self.rank == root:
            buf.copy_(torch.frombuffer(bytearray(data), dtype=torch.uint8))
        self.dist.broadcast(buf, src=root)
        return bytes(buf.cpu().numpy().tobytes())

    def gather_varlen(self, counts, records, total):
        """SURVEY 8(e) collective (3): per-frame variable-length result lists (gs_rect / gs_keypoint records).
        counts: 1-D integer tensor, this rank's per-frame list lengths (frame_range order); records: 2-D tensor
        [cap_per_frame * frames or more, k] or 3-D [frames, cap, k] holding frame f's list in its first counts[f]
        rows.  Returns (counts_all, records_all): the `total` counts in global frame order and the lists
        concatenated in that order, on every rank.  Lists are packed before they travel, so the exchange
        carries sum(counts) records per rank (padded to the largest rank), not frames x cap."""
        import torch
        nloc = counts.numel()
        c = counts.to(torch.int64)
        if nloc == 0:  # a rank that owns no frame (world > total): an empty [0, 1, k] list still takes part in the collectives
            rec = records.new_zeros((0, 1, records.shape[-1]))
        else:
            rec = records if records.dim() == 3 else records.reshape(nloc, -1, records.shape[-1])
        cap, k = rec.shape[1], rec.shape[2]
        keep = torch.arange(cap, device=rec.device)[None, :] < c.to(rec.device)[:, None]
        packed = rec[keep]                                     # [sum(counts), k], frame order kept
        counts_all = self.all_gather_frames(c.to(rec.device), total).to(torch.int64)
        if not self.dist:
            return counts_all, packed
        mine = int(packed.shape[0])
        most = int(self.max_over_ranks(mine))
        pad = torch.zeros((max(most, 1), k), dtype=rec.dtype, device=self._dev())
        pad[:mine] = packed.to(pad.device)
        out = [torch.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(out, pad)
        parts = []
        for r in range(self.world):
            lo, hi = frame_range(r, self.world, total)
            parts.append(out[r][:int(counts_all[lo:hi].sum())])
        return counts_all, torch.cat(parts).to(rec.device)

    def close(self):
        if self.dist and self.dist.is_initialized():
            self.dist.destroy_process_group()
