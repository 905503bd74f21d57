"""GPU: the C-ABI of the relative-BA engine (include/ssk_rba.h, libssk_facade.so) -- define_new_keyframe over a synthetic
trajectory fed as plain observation records: edge creation, landmark bookkeeping, local-area optimisation on the device,
spanning-tree poses, the global pose-graph problem."""
import ctypes as C

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
K = synth.KITTI


def keyframe_observations(ctx, capi, k, P, ids, step=0.54, noise=0.2, rng=None):
    """Stereo observations of world points P (camera frame of keyframe 0: x right, y down, z forward) from the camera moved
    k*step to the right; pixels through ssk_build_observations (projectMatchTo3D + camera pose, as the estimator marshals them)."""
    X = P[:, 0] - k * step
    xl = K["cx"] + K["fx"] * X / P[:, 2]
    yl = K["cy"] + K["fy"] * P[:, 1] / P[:, 2]
    xr = K["cx"] + K["fx"] * (X - K["baseline"]) / P[:, 2]
    vis = (xl > 40) & (xl < synth.W - 40) & (xr > 40) & (yl > 40) & (yl < synth.H - 40)
    n = int(vis.sum())
    kl = np.zeros(n, capi.KEYPOINT_DTYPE); kr = np.zeros(n, capi.KEYPOINT_DTYPE); m = np.zeros(n, capi.DMATCH_DTYPE)
    e = rng.normal(0, noise, (n, 3)) if rng is not None else np.zeros((n, 3))
    kl["x"], kl["y"] = xl[vis] + e[:, 0], yl[vis] + e[:, 1]
    kr["x"], kr["y"] = xr[vis] + e[:, 2], yl[vis] + e[:, 1]
    m["queryIdx"] = m["trainIdx"] = np.arange(n)
    lp = capi.LmParams()
    import synth_lm as sl
    C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
    px, pos = ctx.build_observations(kl, kr, m, lp)
    obs = np.zeros(n, capi.RBA_OBS_DTYPE)
    obs["feat_id"] = ids[vis]
    obs["left_x"], obs["left_y"], obs["right_x"], obs["right_y"] = px[:, 0], px[:, 1], px[:, 2], px[:, 3]
    obs["rel_pos"] = pos
    return obs


def test_rba_engine_sequence(pkg, ctx):
    from ssk_amd import capi
    rng = np.random.default_rng(3)
    P = np.stack([rng.uniform(-15, 40, 900), rng.uniform(-3, 3, 900), rng.uniform(6, 40, 900)], 1)
    ids = np.arange(900, dtype=np.uint64) + 1000
    rba = capi.Rba(ctx, capi.RbaOptions.kitti(submap_size=4, max_tree_depth=3, max_optimize_depth=3))
    n_kf = 11
    infos = []
    for k in range(n_kf):
        obs = keyframe_observations(ctx, capi, k, P, ids, rng=rng)
        assert len(obs) > 150
        infos.append(rba.define_kf(obs, optimize=k > 0))
    assert [i.kf_id for i in infos] == list(range(n_kf))
    assert infos[0].n_created_edges == 0 and all(i.n_created_edges >= 1 for i in infos[1:])
    for i in infos[1:]:
        assert i.total_sqr_error_final <= i.total_sqr_error_init and i.num_iters >= 1 and i.num_observations > 150
        assert i.stg1_obs_rmse < 1.0                                    # 0.2 px noise: sub-pixel after the edge-only stage
    nk, ne, nl = rba.counts()
    assert nk == n_kf and ne >= n_kf - 1 and 300 < nl <= 900
    # fixed-size submaps (SRBA.h:144-308): keyframes 1..3 hang on centre 0, keyframe 4 opens the next local map
    assert [rba.edge(e)[:2] for e in range(3)] == [(0, 1), (0, 2), (0, 3)] and rba.edge(3)[:2] == (0, 4)
    # the camera moved k baselines to its right = -y of the robot frame (camera_pose_on_robot of the ini)
    for k in (1, 5, 10):
        p = rba.spanning_tree_pose(0, k).reshape(3, 4)
        assert abs(p[1, 3] + 0.54 * k) < 0.02 and abs(p[0, 3]) < 0.03 and abs(p[2, 3]) < 0.03, (k, p[:, 3])
        assert np.abs(p[:, :3] - np.eye(3)).max() < 5e-3
    assert rba.spanning_tree_pose(0, 10, max_depth=1) is None           # three edges away
    base, pos = rba.landmark(int(infos[0].kf_id) + 1000 + 5) or (None, None)
    seen0 = keyframe_observations(ctx, capi, 0, P, ids)["feat_id"]
    b, pos = rba.landmark(int(seen0[0]))
    assert b == 0 and np.isfinite(pos).all() and rba.landmark(5) is None
    # the global pose-graph problem: one constraint per edge, nodes = spanning-tree composition; a tree: nothing to optimise
    X, ij, z = rba.global_poses(0)
    assert X.shape == (n_kf, 12) and len(ij) == ne
    Xo, r = ctx.posegraph_optimize(X, ij, z)
    assert r.cost_init < 1e-18 or ne > n_kf - 1
    assert abs(X[10].reshape(3, 4)[1, 3] + 5.4) < 0.02
    rba.close()


def test_rba_engine_errors(pkg, ctx):
    from ssk_amd import capi
    rba = capi.Rba(ctx, capi.RbaOptions.kitti())
    with pytest.raises(capi.SskError):
        rba.edge(0)
    with pytest.raises(capi.SskError):
        capi.Rba(ctx, capi.RbaOptions.kitti(submap_size=1))
    rng = np.random.default_rng(1)
    P = np.stack([rng.uniform(-5, 10, 300), rng.uniform(-3, 3, 300), rng.uniform(6, 30, 300)], 1)
    ids = np.arange(300, dtype=np.uint64)
    rba.define_kf(keyframe_observations(ctx, capi, 0, P, ids), optimize=False)
    # a second keyframe that shares nothing with the map: its single edge is unconstrained, the solve does nothing harmful
    other = keyframe_observations(ctx, capi, 0, P + np.array([0, 0, 3.0]), ids + 5000)
    info = rba.define_kf(other)
    assert info.kf_id == 1 and info.n_created_edges == 1 and info.total_sqr_error_final <= info.total_sqr_error_init + 1e-9
    rba.close()


def test_rba_lock_step_batch_equals_independent_engines(pkg, ctx, monkeypatch):
    """Round 5: ssk_rba_define_kf_batch -- B streams advanced in lock-step, stage 1 of all of them ONE LM batch, the local areas another -- leaves
    every engine in the state B independent ssk_rba_define_kf sequences leave it in: same edges, same keyframe infos, the same bits in every edge
    pose and landmark (the windows are the same and the LM kernels' results do not depend on the batch around a window; SSK_LM_FEW_SPLIT=0 keeps
    a single window on the kernel family its batch uses)."""
    from ssk_amd import capi
    monkeypatch.setenv("SSK_LM_FEW_SPLIT", "0")
    B, n_kf = 12, 9
    opts = capi.RbaOptions.kitti(submap_size=4, max_tree_depth=3, max_optimize_depth=2)
    streams = []
    for b in range(B):
        rng = np.random.default_rng(100 + b)
        P = np.stack([rng.uniform(-15, 40, 700), rng.uniform(-3, 3, 700), rng.uniform(6, 40, 700)], 1)
        ids = np.arange(700, dtype=np.uint64) + 1000 * (b + 1)
        streams.append([keyframe_observations(ctx, capi, k, P, ids, rng=rng) for k in range(n_kf)])
    alone = [capi.Rba(ctx, opts) for _ in range(B)]
    infos_a = [[alone[b].define_kf(streams[b][k], optimize=k > 0) for k in range(n_kf)] for b in range(B)]
    batch = [capi.Rba(ctx, opts) for _ in range(B)]
    tms = []
    infos_b = [[] for _ in range(B)]
    for k in range(n_kf):
        inf, tm = capi.rba_define_kf_batch(batch, [streams[b][k] for b in range(B)], optimize=k > 0)
        tms.append(tm)
        for b in range(B):
            infos_b[b].append(inf[b])
    assert tms[-1]["n_stage1"] == B and tms[-1]["n_stage2"] == B and tms[-1]["stage1_kernel"] == "lm_pose_only_kernel"
    assert tms[-1]["stage2_kernel"].startswith("lm_solve_node") and tms[-1]["stage2_mean_edges"] >= 3
    for b in range(B):
        assert alone[b].counts() == batch[b].counts()
        for ia, ib in zip(infos_a[b], infos_b[b]):
            assert ia.kf_id == ib.kf_id and ia.n_created_edges == ib.n_created_edges and list(ia.created_edge_ids) == list(ib.created_edge_ids)
            assert ia.num_iters == ib.num_iters and ia.num_observations == ib.num_observations
            assert ia.total_sqr_error_init == ib.total_sqr_error_init and ia.total_sqr_error_final == ib.total_sqr_error_final and ia.stg1_obs_rmse == ib.stg1_obs_rmse
        for e in range(alone[b].counts()[1]):
            fa, ta, pa = alone[b].edge(e); fb, tb, pb = batch[b].edge(e)
            assert (fa, ta) == (fb, tb) and np.array_equal(pa, pb), (b, e)
        for fid in streams[b][0]["feat_id"][:40]:
            la, lb = alone[b].landmark(int(fid)), batch[b].landmark(int(fid))
            assert la[0] == lb[0] and np.array_equal(la[1], lb[1])
    for r in alone + batch:
        r.close()
    print("lock-step batch of %d streams, keyframe %d: %s" % (B, n_kf - 1, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tms[-1].items()}))


def test_rba_lock_step_batch_reports_a_failed_engine_and_finishes_the_others(pkg, ctx, monkeypatch):
    """Round 6: an engine whose bookkeeping fails -- here: the keyframe that opens a new local map shares nothing with the map, "the node
    becomes isolated" (SRBA.h:304) -- is named by index in the error text, and the batch goes on for the other engines: they end the keyframe in the
    state independent engines reach (round 5: the first failure aborted the call after every engine had taken its keyframe, none was optimised)."""
    from ssk_amd import capi
    monkeypatch.setenv("SSK_LM_FEW_SPLIT", "0")
    B, n_kf, bad = 5, 5, 2
    opts = capi.RbaOptions.kitti(submap_size=4, max_tree_depth=3, max_optimize_depth=2)
    streams = []
    for b in range(B):
        rng = np.random.default_rng(300 + b)
        P = np.stack([rng.uniform(-15, 40, 600), rng.uniform(-3, 3, 600), rng.uniform(6, 40, 600)], 1)
        ids = np.arange(600, dtype=np.uint64) + 1000 * (b + 1)
        streams.append([keyframe_observations(ctx, capi, k, P, ids, rng=rng) for k in range(n_kf)])
    # keyframe 4 (submap_size 4: the first keyframe of the next local map) of stream `bad`: features nobody has seen
    odd = streams[bad][4].copy(); odd["feat_id"] += 900000
    alone = [capi.Rba(ctx, opts) for _ in range(B)]
    for b in range(B):
        for k in range(n_kf):
            if b == bad and k == 4:
                with pytest.raises(capi.SskError):
                    alone[b].define_kf(odd)
            else:
                alone[b].define_kf(streams[b][k], optimize=k > 0)
    batch = [capi.Rba(ctx, opts) for _ in range(B)]
    for k in range(n_kf):
        lists = [odd if (b == bad and k == 4) else streams[b][k] for b in range(B)]
        if k == 4:
            with pytest.raises(capi.SskError) as ei:
                capi.rba_define_kf_batch(batch, lists, optimize=True)
            msg = str(ei.value)
            assert "engine %d:" % bad in msg and "isolated" in msg and all("engine %d:" % b not in msg for b in range(B) if b != bad), msg
        else:
            capi.rba_define_kf_batch(batch, lists, optimize=k > 0)
    for b in range(B):
        if b == bad:
            continue
        assert alone[b].counts() == batch[b].counts()
        for e in range(alone[b].counts()[1]):
            fa, ta, pa = alone[b].edge(e); fb, tb, pb = batch[b].edge(e)
            assert (fa, ta) == (fb, tb) and np.array_equal(pa, pb), (b, e)
    for r in alone + batch:
        r.close()
