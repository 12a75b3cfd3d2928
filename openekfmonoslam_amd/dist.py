"""One-process-per-GPU plumbing for bench.py and the multi-GPU path: rendezvous, barrier, max-over-ranks timing and
the row partition of the covariance.  torch.distributed is the transport (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests); nothing here computes filter arithmetic."""
import os

import numpy as np


class Ranks:
    """Thin wrapper so single-process runs need no process group at all."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist

            # EKF_DIST_BACKEND=gloo lets several ranks share one GPU (functional checks on a 1-GPU box; RCCL refuses that)
            backend = backend or os.environ.get("EKF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if torch.cuda.is_available():
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group(backend, device_id=self.device)
            else:
                dist.init_process_group(backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        """max of a python float over all ranks (the contract's timing reduction)."""
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def all_gather_rows(self, local_rows, total_rows):
        """All-gather of row blocks of unequal height (the exchange step of the sharded update: every rank ends up
        with all rows of B).  local_rows: [rows_r, cols] float array of this rank; returns [total_rows, cols]."""
        if self.dist is None:
            return local_rows
        import torch

        cols = local_rows.shape[1]
        counts = [None] * self.world
        self.dist.all_gather_object(counts, int(local_rows.shape[0]))
        assert sum(counts) == total_rows, (counts, total_rows)
        mx = max(counts)
        pad = np.zeros((mx, cols), dtype=local_rows.dtype)
        pad[: local_rows.shape[0]] = local_rows
        src = torch.from_numpy(pad)
        if self.device is not None:
            src = src.to(self.device)
        bufs = [torch.empty_like(src) for _ in range(self.world)]
        self.dist.all_gather(bufs, src)
        return np.concatenate([b.cpu().numpy()[:c] for b, c in zip(bufs, counts)], axis=0)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def shard_rows(n_features, world, rank):
    """Row block [lo, hi) of P owned by `rank` -- the same arithmetic as ekf_shard_rows in the C ABI (camera rows
    on rank 0, features split contiguously, sizes differing by at most one feature)."""
    per, extra = divmod(n_features, world)
    f0 = rank * per + min(rank, extra)
    f1 = f0 + per + (1 if rank < extra else 0)
    return (0 if rank == 0 else 13 + 6 * f0), 13 + 6 * f1
