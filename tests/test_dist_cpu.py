"""CPU (gloo, world size 2): the N>1 plumbing of the path -- partitioning and the single all-gather of the
shared-landmark blocks with its rank-major layout; the fusion (numpy restatement of ssk_lm_stitch_shared's formula,
test infrastructure) must be bit-identical on both ranks because it sums in rank order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fuse_numpy(g, n_ranks, n_shared):
    """x = (sum_r H_r + eps I)^-1 sum_r H_r x_r, summed in rank order (what stitch_shared_kernel computes)."""
    g = g.reshape(n_ranks, n_shared, 12)
    out = np.zeros((n_shared, 3))
    for s in range(n_shared):
        A = np.zeros((3, 3))
        b = np.zeros(3)
        for r in range(n_ranks):
            p = g[r, s]
            Hm = np.array([[p[6], p[7], p[8]], [p[7], p[9], p[10]], [p[8], p[10], p[11]]])
            A = A + Hm
            b = b + Hm @ p[:3]
        out[s] = np.linalg.solve(A + 1e-9 * np.eye(3), b)
    return out


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ssk_dist", os.path.join(ROOT, "srba-stereo-slam_amd", "dist.py"))
    d = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(d)
    r, w = d.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n_shared = 16
    rng = np.random.default_rng(100 + rank)
    blocks = np.zeros((n_shared, 12))
    blocks[:, :3] = rng.normal(0, 5, (n_shared, 3)) + np.arange(n_shared)[:, None]
    for s in range(n_shared):
        J = rng.normal(0, 1, (4, 3))
        Hm = J.T @ J + 0.1 * np.eye(3)
        blocks[s, 6:] = [Hm[0, 0], Hm[0, 1], Hm[0, 2], Hm[1, 1], Hm[1, 2], Hm[2, 2]]
    shared = torch.from_numpy(blocks.reshape(-1).copy())
    gathered = torch.zeros(world * n_shared * 12, dtype=torch.float64)
    d.exchange_shared(shared, gathered)
    g = gathered.numpy().reshape(world, n_shared, 12)
    assert np.array_equal(g[rank], blocks)                       # rank-major layout
    fused = fuse_numpy(gathered.numpy(), world, n_shared)
    parts = [d.partition(11, k, world) for k in range(world)]
    q.put((rank, fused.tobytes(), g[1 - rank][:, :3].tobytes(), parts))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_exchange_is_rank_ordered_and_deterministic():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=120)
        res[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                                # fused landmarks bit-identical on both ranks
    assert res[0][3] == [(0, 6), (6, 11)]                        # contiguous partition, remainder to the low ranks
    # each rank really received the OTHER rank's positions
    a = np.frombuffer(res[0][2]).reshape(-1, 3)
    b = np.frombuffer(res[1][2]).reshape(-1, 3)
    assert not np.array_equal(a, b)
