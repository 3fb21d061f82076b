"""Worker of test_sharded_pass_gloo_world2: run with torch.distributed.run, 2 ranks, gloo."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fastlivo  # noqa: E402,F401
from fast_livo_amd import synth  # noqa: E402
from fast_livo_amd.sharded import ShardedPass, shard_range  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import p  # noqa: E402


class EmulLioBackend:
    """Same protocol as sharded.GpuLioBackend, arithmetic from tests/host_emul (test-only)."""

    def __init__(self, E, fr, body, plane, sel):
        self.E, self.fr, self.body, self.plane, self.sel = E, fr, body, plane, sel
        self.x = orc.state18_from_frame(fr).vec().copy()
        self.xp = self.x.copy()
        self.delta = np.zeros(18)

    def accumulate(self):
        s = np.zeros(32)
        RLI = np.ascontiguousarray(self.fr.R_LI.reshape(9))
        tLI = np.ascontiguousarray(self.fr.t_LI)
        self.E.emul_lio18_accumulate(p(self.body, C.c_float), p(self.plane, C.c_float), p(self.sel, C.c_uint8), self.body.shape[0],
                                     p(self.x, C.c_double), p(RLI, C.c_double), p(tLI, C.c_double), p(s, C.c_double), None)
        return torch.from_numpy(s)

    def solve(self, rec):
        s = rec.numpy().copy()
        P = np.ascontiguousarray(self.fr.cov18.reshape(-1))
        G6 = np.zeros(108)
        self.E.emul_solve18_fast(p(self.x, C.c_double), p(self.xp, C.c_double), p(P, C.c_double), C.c_double(self.fr.laser_point_cov),
                                 p(s, C.c_double), C.c_double(1.0), p(G6, C.c_double), p(self.delta, C.c_double))


class EmulVioChunkBackend:
    """The VIO side of ShardedPass's protocol (accumulate / chunk / gather_buffer / solve(rec, gathered, world)) around the oracle's
    per-patch errors: checks what the orchestration must deliver to fl_vio_solve_exact -- every rank's chunk, in rank order, at the
    agreed stride -- and takes the accept decision on the reference's float chain over the gathered floats."""

    def __init__(self, errors_mine, stride, errors_all, n_meas):
        self.e, self.stride, self.all, self.n = errors_mine, stride, errors_all, n_meas
        self.decision = None
        self._all = None

    def accumulate(self):
        rec = np.zeros(32)
        rec[28] = float(np.sum(self.e.astype(np.float64)))
        rec[27] = 64.0 * len(self.e)
        return torch.from_numpy(rec)

    def chunk(self):
        c = np.zeros(self.stride, np.float32)
        c[:1].view(np.int32)[0] = len(self.e)
        c[1:1 + len(self.e)] = self.e
        return torch.from_numpy(c)

    def gather_buffer(self, world):
        if self._all is None:
            self._all = torch.zeros(world * self.stride, dtype=torch.float32)
        return self._all

    def solve(self, rec, gathered, world):
        g = gathered.numpy()
        parts = []
        for r in range(world):
            seg = g[r * self.stride:(r + 1) * self.stride]
            m = int(seg[:1].view(np.int32)[0])
            parts.append(seg[1:1 + m])
        flat = np.concatenate(parts)
        assert np.array_equal(flat.view(np.uint32), self.all.view(np.uint32)), "gathered per-patch floats are not the frame's, in order"
        f = np.float32(0.0)
        for e in flat:                       # lidar_selection.cpp:851-852: error += patch_error, float
            f = np.float32(f + e)
        assert int(rec[27].item()) == self.n
        self.decision = np.float32(f / np.float32(self.n))


def vio_part(rank, world):
    """the chunk protocol of the exact VIO accept test on gloo (sharded.py: agree_stride, all_gather_into_tensor in rank order)"""
    from fast_livo_amd.sharded import agree_stride
    scene = synth.make_scene()
    fr = synth.make_lio_frame(400, scene=scene)
    vf = synth.make_vio_frame(301, fr)               # odd: the ranks hold different patch counts
    vf.max_iterations = 1
    x = orc.state18_from_frame(fr)
    r = orc.vio_update_state(vf, x, x.copy(), 1e10, 0)
    lo, hi = shard_range(vf.m, rank, world)
    stride = agree_stride(hi - lo, dist)
    assert stride == max(shard_range(vf.m, q, world)[1] - shard_range(vf.m, q, world)[0] for q in range(world)) + 1
    be = EmulVioChunkBackend(r["errors"][lo:hi].copy(), stride, r["errors"], 64 * vf.m)
    ShardedPass(be, dist).step()
    assert be.decision == np.float32(r["error"]), (be.decision, r["error"])       # the reference's `error`, bit for bit, on every rank


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    vio_part(rank, world)
    E = C.CDLL(os.path.join(ROOT, "tests", "host_emul", "libemul.so"))
    scene = synth.make_scene()
    n = 6000
    fr = synth.make_lio_frame(n, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    lo, hi = shard_range(n, rank, world)
    body = np.ascontiguousarray(fr.body_xyz[lo:hi])
    nb = np.ascontiguousarray(nbr[lo:hi])
    plane = np.zeros((hi - lo, 4), np.float32)
    ok = np.zeros(hi - lo, np.uint8)
    E.emul_fit_planes(p(nb, C.c_float), hi - lo, p(plane, C.c_float), p(ok, C.c_uint8))
    sel = (valid[lo:hi] & ok).astype(np.uint8)
    be = EmulLioBackend(E, fr, body, plane, sel)
    sp = ShardedPass(be, dist)
    # unsharded oracle, three passes without re-search
    xo = orc.state18_from_frame(fr)
    xpo = xo.copy()
    sel_o = valid.copy()
    G = np.zeros((18, 18))
    nvo = np.zeros((n, 4), np.float32)
    rl = np.zeros(n)
    for it in range(3):
        ro = orc.lio18_iterate(xo, xpo, fr.body_xyz, nbr, sel_o, fr.R_LI, fr.t_LI, fr.laser_point_cov, G=G, normvec=nvo, res_last=rl)
        rec = sp.step()
        assert int(rec[27].item()) == ro["out"].effct_feat_num, (it, rec[27].item(), ro["out"].effct_feat_num)
        assert np.abs(be.delta - np.array(ro["out"].solution)).max() <= 1e-9
        assert np.abs(be.x - xo.vec()).max() <= 1e-9
        # every rank holds the same state bit for bit (identical reduced record -> identical solve)
        t = torch.from_numpy(be.x.copy())
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert all(torch.equal(gathered[0], g) for g in gathered)
    print(f"RANK{rank} OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
