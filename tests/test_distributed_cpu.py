"""world_size-2 `gloo` tests of the N>1 plumbing (no GPU): frame sharding, the MAX time reduction bench.py uses,
the landmark sharding of a BA problem and the all-reduce callback the C library calls back into."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stella_vslam_amd import distributed as D
from stella_vslam_amd import synthetic as S


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. time reduction
        assert D.max_over_ranks(1.0 + rank) == float(world)
        # 2. frame shards are a partition
        mine = list(D.frame_shard(11, rank, world))
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        assert sorted(sum(allv, [])) == list(range(11))
        # 3. the C-callable all-reduce sums host doubles in place (gloo path of the callback)
        cb, keep = D.make_allreduce_callback(host_buffers=True)
        buf = np.arange(5, dtype=np.float64) * (rank + 1)
        rc = cb(None, buf.ctypes.data, buf.size, None)
        assert rc == 0 and np.array_equal(buf, np.arange(5) * sum(range(1, world + 1)))
        # 4. landmark shards partition the observations and keep landmarks whole
        sc = S.ba_scene(num_kf=5, num_lm=40, obs_per_lm=3, num_fixed=2, seed=1)
        sh = D.shard_by_landmark(sc, rank, world)
        assert (sh["obs_point"] % world == rank).all() and len(sh["pose_cw"]) == 5 and len(sh["points"]) == 40
        counts = [None] * world
        dist.all_gather_object(counts, sh["_obs_index"].tolist())
        assert sorted(sum(counts, [])) == list(range(len(sc["obs_pose"])))
        # 5. the DISTRIBUTED FACTORISATION of the reduced camera system (csrc/ba_skyline.hip: segmented envelope elimination): this rank
        #    eliminates only the jobs it owns, what they leave on the separators and the solution cross ranks through this very all-reduce
        #    (host walk of the plan the kernels walk: svgpu_selftest_segmented_solve_rank) -- every rank ends with the single-rank solution, bit for bit
        from stella_vslam_amd._lib import lib
        from tests.test_sky_segments import _ring, _system
        n = 300
        ab, Sb, g, dense = _system(n, _ring(n, 4), seed=3)
        x1, xr = np.zeros(6 * n), np.zeros(6 * n)
        info1, infor = np.zeros(8, np.int32), np.zeros(8, np.int32)
        p = lambda a: C.c_void_p(a.ctypes.data)
        assert lib().svgpu_selftest_segmented_solve(n, len(ab), p(ab), p(Sb), p(g), 5, 1, p(x1), p(info1)) == 0
        rc = lib().svgpu_selftest_segmented_solve_rank(n, len(ab), p(ab), p(Sb), p(g), 5, rank, world, cb, None, p(xr), p(infor))
        assert rc == 0 and np.array_equal(xr, x1), (rc, np.abs(xr - x1).max())
        assert 0 < infor[5] < infor[2]  # a strict subset of the jobs ran here
        owned = [None] * world
        dist.all_gather_object(owned, int(infor[5]))
        assert sum(owned) == int(infor[2])
        assert np.abs(x1 - np.linalg.solve(dense, g)).max() < 1e-9
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_frame_shard_balanced():
    for n, w in ((64, 8), (10, 3), (2, 4), (0, 2)):
        parts = [D.frame_shard(n, r, w) for r in range(w)]
        assert sum(len(p) for p in parts) == n
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        assert [i for p in parts for i in p] == list(range(n))
