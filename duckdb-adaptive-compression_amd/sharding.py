"""Multi-GPU sharding of column segments: one process per GPU, segments partitioned by segment id, per-GPU
segment pools, NO data-path collective (segments are independent: own min, width, count, words — SURVEY.md
§8e).  torch.distributed is used only for the rendezvous, the timing barrier and the scalar reductions of the
measurement (max elapsed time, total rows); backend "nccl" (= RCCL) on GPUs, "gloo" in the CPU tests."""
import os


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when launched directly."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def segment_range(nseg_global, rank, world):
    """Contiguous range [lo, hi) of global segment ids owned by `rank`: balanced to within one segment."""
    lo = nseg_global * rank // world
    hi = nseg_global * (rank + 1) // world
    return lo, hi


def owner_of(seg_id, nseg_global, world):
    """Inverse of segment_range."""
    r = min(world - 1, (seg_id * world + world - 1) // max(nseg_global, 1))
    while r > 0 and seg_id < nseg_global * r // world:
        r -= 1
    while r < world - 1 and seg_id >= nseg_global * (r + 1) // world:
        r += 1
    return r


def column_shard(total_rows, type_size, rank, world):
    """The slice of ONE global column that `rank` owns: the column's Appender segment list (layout.py) is partitioned
    by segment id into `world` contiguous ranges (segment_range); returns (seg_lo, seg_hi, row_lo, row_hi, counts)
    where counts are the row counts of the rank's segments and [row_lo, row_hi) their rows in the global column.
    The reference's unit of independence is the row group / segment (row_group_collection.cpp:119-155)."""
    import numpy as np
    from .layout import appender_segment_counts
    counts = appender_segment_counts(total_rows, type_size)
    lo, hi = segment_range(len(counts), rank, world)
    starts = np.concatenate([[0], np.cumsum(counts.astype(np.uint64))])
    return lo, hi, int(starts[lo]), int(starts[hi]), counts[lo:hi].copy()


class Comm:
    """Rendezvous + the three scalar collectives the measurement needs."""

    def __init__(self, backend=None, device=None):
        self.rank, self.local_rank, self.world = dist_env()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = torch.device("cuda", device)
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend

    def _tensor(self, x, dtype):
        import torch
        dev = "cpu"
        if self.dist is not None and self.backend == "nccl":
            dev = "cuda:%d" % self.device
        return torch.tensor([x], dtype=dtype, device=dev)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = self._tensor(float(x), torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, x):
        if self.dist is None:
            return int(x)
        import torch
        t = self._tensor(int(x), torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather_rows(self, row):
        """One short list of numbers per rank -> the list of every rank's list, on every rank (all_gather of a small
        float64 tensor: rows and byte counts stay exact below 2^53)."""
        row = [float(x) for x in row]
        if self.dist is None:
            return [row]
        import torch
        dev = "cuda:%d" % self.device if self.backend == "nccl" else "cpu"
        mine = torch.tensor(row, dtype=torch.float64, device=dev)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [t.cpu().tolist() for t in out]

    def sum_u64(self, x):
        """Sum mod 2^64 of one unsigned 64-bit value per rank (checksum of checksums): reduced as two 32-bit halves
        so that no backend's integer overflow behaviour matters."""
        x &= 0xFFFFFFFFFFFFFFFF
        lo = self.sum(x & 0xFFFFFFFF)
        hi = self.sum(x >> 32)
        return (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
