"""Point/patch-range sharding of one ESKF pass across ranks.

Every rank owns a contiguous range of the scan points (and patches), reduces its partial normal
equations on its device, the ranks all-reduce (sum) the small fp64 record and each rank runs the
gain solve redundantly on bitwise-identical inputs, so all ranks hold the same new state and no
broadcast is needed.  The collective is `torch.distributed.all_reduce` -- RCCL over xGMI with the
"nccl" backend on MI355X, gloo in the CPU tests.

The compute backend is pluggable so that the orchestration (ranges, collective, ordering) is the
same code on the GPU (capi.Handle) and in the world_size-2 gloo test (tests/host_emul arithmetic).
"""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int):
    """Contiguous range [lo, hi) of rank `rank`: [r*n/G, (r+1)*n/G)."""
    lo = (rank * n) // world
    hi = ((rank + 1) * n) // world
    return lo, hi


class ShardedPass:
    """One sharded ESKF pass: accumulate -> all_reduce(sum) -> solve.

    backend.accumulate() must return a 1-D float64 torch tensor (the reduction record) living where
    the process group can reduce it; backend.solve(record) consumes the reduced record.
    """

    def __init__(self, backend, dist=None):
        self.backend = backend
        self.dist = dist

    def step(self):
        rec = self.backend.accumulate()
        live = self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1
        if live:
            self.dist.all_reduce(rec)          # default op = SUM
        # VIO: the accept test is the reference's FLOAT running sum over the patches in order (lidar_selection.cpp:849-861). A backend
        # that exports its per-patch floats (`chunk()`: [count, floats, padding], equal size on every rank) gets them all-gathered
        # in rank order, and its solve replays the chain over all ranks' patches itself -- bit-identical decisions on all ranks.
        chunk = self.backend.chunk() if hasattr(self.backend, "chunk") else None
        if chunk is not None:
            world = self.dist.get_world_size() if live else 1
            gathered = self.backend.gather_buffer(world)
            if live:
                self.dist.all_gather_into_tensor(gathered, chunk)
            else:
                gathered.copy_(chunk)
            self.backend.solve(rec, gathered, world)
        else:
            self.backend.solve(rec)
        return rec


def agree_stride(count, dist=None, device="cpu"):
    """Chunk size of the per-patch gather: the largest patch count of any rank + 1 (one max all-reduce, once per patch set)."""
    import torch
    t = torch.tensor([int(count) + 1], dtype=torch.int64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


class GpuLioBackend:
    """capi.Handle-backed Mode-18 LIO pass for ShardedPass."""

    def __init__(self, handle, flags):
        import torch
        from . import capi
        self.h = handle
        self.flags = flags
        self.buf = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")

    def accumulate(self):
        self.h.lio_accumulate18(self.buf.data_ptr(), self.flags)
        return self.buf

    def solve(self, rec):
        self.h.lio_solve18(rec.data_ptr(), self.flags)


class GpuVioBackend:
    """capi.Handle-backed VIO pass (one pyramid level) for ShardedPass. stride (agree_stride over the ranks' patch counts): the pass
    decides its accept test on the reference's float chain over all ranks' patches (fl_vio_solve_exact); None: the fp64 comparison
    of fl_vio_solve (forced benchmark passes, which take no decision)."""

    def __init__(self, handle, level, flags, stride=None):
        import torch
        from . import capi
        self.h = handle
        self.level = level
        self.flags = flags
        self.stride = stride
        self.buf = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")
        self._chunk = torch.zeros(stride, dtype=torch.float32, device="cuda") if stride else None
        self._all = None

    def accumulate(self):
        self.h.vio_accumulate(self.level, self.buf.data_ptr())
        return self.buf

    def chunk(self):
        if self._chunk is None:
            return None
        self.h.vio_errors_chunk(self._chunk.data_ptr(), self.stride)
        return self._chunk

    def gather_buffer(self, world):
        import torch
        if self._all is None or self._all.numel() != world * self.stride:
            self._all = torch.zeros(world * self.stride, dtype=torch.float32, device="cuda")
        return self._all

    def solve(self, rec, gathered=None, world=1):
        if gathered is None:
            self.h.vio_solve(rec.data_ptr(), self.flags)
        else:
            self.h.vio_solve_exact(rec.data_ptr(), self.flags, gathered.data_ptr(), self.stride, world)
