"""Row-sharded filter across GPUs: one EkfEngine per rank, each holding the camera rows of P and the
rows of the features it owns; the one exchange per prediction (all-gather of the H.P row blocks and of the 2x2 S_i)
is carried by torch.distributed (backend "nccl" = RCCL over xGMI, one process per GPU) or, when the ranks share one
GPU (the single-GPU emulation the tests run), by device-to-device copies between the engines' buffers.

Nothing here computes filter arithmetic; the engines do, through the C ABI."""
import ctypes as C
import threading

import numpy as np

from . import engine as _engine


def exchange_rows(dist, buf_u8, row_bytes, row_begin, rank):
    """In-place all-gather of row blocks of unequal height: rank r owns rows [row_begin[r], row_begin[r+1]) of
    `buf_u8` (a flat uint8 torch tensor: the table, row_bytes per row) and receives everybody else's.  One broadcast
    per owner: blocks differ in size by at most one feature, and over point-to-point xGMI a broadcast of block r is
    the same traffic per link as its slot in a ring all-gather."""
    world = len(row_begin) - 1
    for r in range(world):
        lo, hi = row_begin[r] * row_bytes, row_begin[r + 1] * row_bytes
        if hi > lo:
            dist.broadcast(buf_u8[lo:hi], src=r)


class _DevicePtr:
    """Minimal __cuda_array_interface__ carrier so torch can alias an engine buffer without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class DistributedExchange:
    """Exchange callback for one-process-per-GPU runs (torch.distributed already initialised)."""

    def __init__(self, dist, device):
        import torch

        self.torch, self.dist, self.device = torch, dist, device

    def __call__(self, what, base, row_bytes, row_begin, world, rank):
        nbytes = row_begin[-1] * row_bytes
        if nbytes == 0:
            return 0
        t = self.torch.as_tensor(_DevicePtr(base, nbytes), device=self.device)
        exchange_rows(self.dist, t, row_bytes, row_begin, rank)
        self.torch.cuda.synchronize(self.device)
        return 0


class LocalShardGroup:
    """`world` ranks of one sharded filter inside ONE process on ONE GPU: every rank is its own engine with its own
    stream and row block; each runs in its own thread (ctypes releases the GIL inside the C calls) and the exchange
    is a barrier + device-to-device pulls.  Same partition logic and the same engine code path as the multi-GPU
    run; only the transport differs."""

    def __init__(self, cam, par, max_features, world, transport="hip", **kw):
        """transport "hip": pulls through ekf_device_copy; "torch": pulls through torch tensors that alias the engine
        buffers (the aliasing DistributedExchange relies on, exercised without a second GPU)."""
        self.world = world
        self.transport = transport
        self.engines = [_engine.EkfEngine(cam, par, max_features, shard=(r, world), **kw) for r in range(world)]
        self._barrier = threading.Barrier(world)
        self._bases = {}
        self.bytes_exchanged = 0
        self.bytes_by_kind = {}  # what rank 0 pulled, per kind of exchange (EKF_XCHG_*)
        for r, e in enumerate(self.engines):
            e.set_exchange(self._make_exchange(r))

    def _make_exchange(self, me):
        def fn(what, base, row_bytes, row_begin, world, rank):
            assert rank == me
            self._bases[(what, rank)] = base
            self._barrier.wait()  # everyone has produced its block and published its base address
            eng = self.engines[me]
            for r in range(world):
                lo, hi = row_begin[r] * row_bytes, row_begin[r + 1] * row_bytes
                if r != me and hi > lo:
                    if self.transport == "torch":
                        import torch

                        dst = torch.as_tensor(_DevicePtr(base + lo, hi - lo), device="cuda")
                        src = torch.as_tensor(_DevicePtr(self._bases[(what, r)] + lo, hi - lo), device="cuda")
                        dst.copy_(src)
                        torch.cuda.synchronize()
                    else:
                        eng.device_copy(base + lo, self._bases[(what, r)] + lo, hi - lo)
                    if me == 0:
                        self.bytes_exchanged += hi - lo
                        self.bytes_by_kind[what] = self.bytes_by_kind.get(what, 0) + hi - lo
            self._barrier.wait()  # nobody overwrites a block somebody is still pulling
            return 0

        return fn

    def run(self, fn):
        """fn(rank, engine) on every rank concurrently; returns the per-rank results (re-raises the first error)."""
        out, err = [None] * self.world, [None] * self.world

        def work(r):
            try:
                out[r] = fn(r, self.engines[r])
            except BaseException as ex:  # noqa: BLE001
                err[r] = ex
                self._barrier.abort()

        th = [threading.Thread(target=work, args=(r,)) for r in range(self.world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for ex in err:
            if ex is not None and not isinstance(ex, threading.BrokenBarrierError):
                raise ex
        for ex in err:
            if ex is not None:
                raise ex
        return out

    def set_state(self, x13, feature_pos, feature_type, desc, P):
        for e in self.engines:
            e.set_state(x13, feature_pos, feature_type, desc, P)

    def get_state(self):
        """state of rank 0 (replicated) and the covariance assembled from every rank's rows"""
        n = self.engines[0].n
        P = np.full((n, n), np.nan)
        x = fp = None
        for e in self.engines:
            x, fp, _ = e.get_state(P_out=P)
        return x, fp, P

    def close(self):
        for e in self.engines:
            e.close()
