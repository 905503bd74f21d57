"""CPU (gloo, world size 2): the N > 1 path of BASELINE configs[4] / SURVEY 8(e) -- partitioning, the single all-gather of the
shared-landmark blocks with its rank-major layout, and what the exchange MEANS: two ranks, one submap each, 24 landmarks of one
identity observed from both.  Per round every rank solves its own window (the CPU oracle stands in for the GPU solve: test
infrastructure), contributes its g / H blocks of the shared landmarks (tests/shared_ref.py, the numpy restatement of
ssk_lm_batch_pack_shared), ONE all-gather, the rank-ordered Gauss-Newton update (restatement of ssk_lm_stitch_shared).  The rounds
must converge to the JOINT single-process solve of the union of the two submaps, bit-identically on both ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SHARED, ROUNDS = 24, 70
KW = dict(n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60, opt_depth=2)


def worker(rank, world, port, q, rounds=ROUNDS, accelerate=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import importlib.util
    spec = importlib.util.spec_from_file_location("ssk_dist", os.path.join(ROOT, "srba-stereo-slam_amd", "dist.py"))
    d = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(d)
    import oracle_lib as ol
    import shared_ref as sr
    import synth_lm as sl
    r, w = d.init_from_env("gloo")
    assert (r, w) == (rank, world)
    oracle = ol.load()
    wins, idxs, sh = sl.make_shared_submaps(n_sub=world, n_shared=N_SHARED, seed0=50, only={rank}, **KW)
    win, idx = wins[0], idxs[0]
    prm = sl.make_params(max_iters=50, conv=1e-14)
    pos = sh["init"].copy()
    errs, first = [], None
    acc = d.SharedLandmarkAccelerator(m=5) if accelerate else None
    for rnd in range(rounds):
        win.lm_pos[idx[idx >= 0]] = pos[idx >= 0]
        out = oracle.lm_solve(win, prm, trace=False)
        win = sl.Window(win.edge_unknown_slot, out["poses"], win.lm_unknown_slot, out["lms"], win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
        blocks = sr.blocks(win, prm, out["poses"], out["lms"], idx)
        shared = torch.from_numpy(blocks.reshape(-1).copy())
        gathered = torch.zeros(world * N_SHARED * 12, dtype=torch.float64)
        d.exchange_shared(shared, gathered)                      # the ONE collective of a round
        g = gathered.numpy().reshape(world, N_SHARED, 12)
        assert np.array_equal(g[rank], blocks)                   # rank-major layout
        if rnd == 0:
            first = g[(rank + 1) % world][:, 3:6].tobytes()      # ANOTHER rank's gradient blocks really arrived
        gx = sr.consensus(g, pos)
        pos = acc.step(pos, gx).reshape(N_SHARED, 3) if acc else gx      # (Anderson mixing of the rounds: the same arithmetic on every rank)
        t = torch.tensor([out["res"].err_final], dtype=torch.float64)
        dist.all_reduce(t)                                       # (test bookkeeping only: the joint cost)
        errs.append(float(t.item()))
    parts = [d.partition(11, k, world) for k in range(world)]
    q.put((rank, pos.tobytes(), win.edge_inv_pose.tobytes(), win.lm_pos.tobytes(), errs, first, parts))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_exchange_rounds_equal_the_joint_solve():
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
        r = q.get(timeout=300)
        res[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                                # consensus positions bit-identical on both ranks (rank-ordered sums)
    assert res[0][6] == [(0, 6), (6, 11)]                        # contiguous partition, remainder to the low ranks
    assert res[0][5] != res[1][5]                                # each rank received the other's blocks, not its own twice
    # ---- against the joint single-process solve of the union ----
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    import synth_lm as sl
    wins, idxs, sh = sl.make_shared_submaps(n_sub=2, n_shared=N_SHARED, seed0=50, **KW)
    joint, e_off, lm_map, sj = sl.join_windows(wins, idxs)
    J = ol.load().lm_solve(joint, sl.make_params(max_iters=50, conv=1e-14), trace=False)
    shared = np.frombuffer(res[0][1]).reshape(N_SHARED, 3)
    assert np.abs(shared - J["lms"][sj]).max() <= 1e-6
    errs = res[0][4]
    assert errs[-1] < errs[0] and abs(errs[-1] - J["res"].err_final) <= 1e-9 * J["res"].err_final
    for r in range(2):
        poses = np.frombuffer(res[r][2]).reshape(-1, 12)
        lms = np.frombuffer(res[r][3]).reshape(-1, 3)
        assert np.abs(poses - J["poses"][e_off[r]:e_off[r] + len(poses)]).max() <= 1e-6, r
        private = idxs[r][idxs[r] >= 0]
        keep = np.ones(len(lms), bool)
        keep[private] = False                                    # (the window's own copy of the shared ones is one round behind)
        ref = J["lms"][lm_map[r]][keep]                          # 1e-6 RELATIVE: a landmark 40 m away is known to its depth's share of that
        assert (np.abs(lms[keep] - ref) <= 1e-6 * np.maximum(1.0, np.linalg.norm(ref, axis=1))[:, None]).all(), r


def test_gloo_world4_accelerated_rounds_equal_the_joint_solve():
    """World size 4 (every shared landmark seen from two of the four submaps) with the Anderson-accelerated rounds of
    srba-stereo-slam_amd/dist.py (SharedLandmarkAccelerator): 20 rounds instead of 70 reach the joint solve of the union of the four
    submaps within 1e-6, bit-identically on all ranks."""
    world, rounds = 4, 20
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q, rounds, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=600)
        res[r[0]] = r
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(res[r][1] == res[0][1] for r in range(world))     # consensus positions bit-identical on every rank
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    import synth_lm as sl
    wins, idxs, sh = sl.make_shared_submaps(n_sub=world, n_shared=N_SHARED, seed0=50, **KW)
    joint, e_off, lm_map, sj = sl.join_windows(wins, idxs)
    J = ol.load().lm_solve(joint, sl.make_params(max_iters=50, conv=1e-14), trace=False)
    shared = np.frombuffer(res[0][1]).reshape(N_SHARED, 3)
    assert np.abs(shared - J["lms"][sj]).max() <= 1e-6
    for r in range(world):
        poses = np.frombuffer(res[r][2]).reshape(-1, 12)
        assert np.abs(poses - J["poses"][e_off[r]:e_off[r] + len(poses)]).max() <= 1e-6, r
