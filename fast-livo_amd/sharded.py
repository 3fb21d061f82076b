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
        if self.dist is not None and self.dist.is_initialized() and self.dist.get_world_size() > 1:
            self.dist.all_reduce(rec)          # default op = SUM
        self.backend.solve(rec)
        return rec


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
    """capi.Handle-backed VIO pass (one pyramid level) for ShardedPass."""

    def __init__(self, handle, level, flags):
        import torch
        from . import capi
        self.h = handle
        self.level = level
        self.flags = flags
        self.buf = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")

    def accumulate(self):
        self.h.vio_accumulate(self.level, self.buf.data_ptr())
        return self.buf

    def solve(self, rec):
        self.h.vio_solve(rec.data_ptr(), self.flags)
