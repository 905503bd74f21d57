"""CPU tests pinning the oracle's relative-BA restatement with known-answer properties (the reference has
no fixtures for this path and SRBA is not in the image):
analytic Jacobians vs finite differences, exact recovery on a noise-free scene, monotone error, the SE(3)
update rule, and a committed golden trace."""
import os

import numpy as np
import pytest

import synth_lm as sl

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def small(seed=1, **kw):
    d = dict(n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    d.update(kw)
    return sl.make_submap_problem(seed, **d)


def test_window_shape():
    win, gt = small()
    assert win.E >= 5 and win.L >= 50 and len(win.obs_lm) > 200
    plen = np.diff(win.obs_path_off)
    assert plen.min() == 0 and 2 <= plen.max() <= 4       # observed from its base / across submaps
    assert set(np.unique(win.path_edge & 1)) == {0, 1}    # both traversal directions occur


@pytest.mark.parametrize("robust", [False, True])
def test_gradient_matches_finite_differences(oracle, robust):
    """g = J^T r must equal -1/2 d(err)/d(eps) for the left-multiplicative edge update and additive landmarks."""
    win, gt = small(2)
    prm = sl.make_params(robust=robust, sensor_t=(0.3, -0.1, 0.2))
    gp, gf = oracle.lm_gradient(win, prm)
    e0 = oracle.lm_total_error(win, prm)
    h = 1e-6
    slot_edge = {int(s): i for i, s in enumerate(win.edge_unknown_slot) if s >= 0}
    rng = np.random.default_rng(0)
    for s in rng.choice(win.E, min(win.E, 4), replace=False):
        for a in range(6):
            eps = np.zeros(6)
            eps[a] = h
            poses = win.edge_inv_pose.copy()
            poses[slot_edge[int(s)]] = oracle.se3_left_update(eps, poses[slot_edge[int(s)]])
            ep = oracle.lm_total_error(win, prm, poses=poses)
            eps[a] = -h
            poses[slot_edge[int(s)]] = oracle.se3_left_update(eps, win.edge_inv_pose[slot_edge[int(s)]])
            em = oracle.lm_total_error(win, prm, poses=poses)
            fd = -(ep - em) / (4 * h)
            assert abs(fd - gp[6 * s + a]) <= 1e-5 * max(1.0, abs(fd)), (s, a, fd, gp[6 * s + a])
    for l in rng.choice(win.L, 6, replace=False):
        for a in range(3):
            lms = win.lm_pos.copy()
            lms[l, a] += h
            ep = oracle.lm_total_error(win, prm, lms=lms)
            lms[l, a] -= 2 * h
            em = oracle.lm_total_error(win, prm, lms=lms)
            fd = -(ep - em) / (4 * h)
            assert abs(fd - gf[3 * l + a]) <= 1e-5 * max(1.0, abs(fd))
    assert e0 > 0


def test_se3_left_update_rule(oracle):
    """L3: exp(eps)(+)T with the pseudo-exponential: R <- Rodrigues(w) R, t <- Rodrigues(w) t + tau."""
    rng = np.random.default_rng(3)
    T = np.hstack([sl.so3_exp(rng.normal(0, 1, 3)), rng.normal(0, 2, (3, 1))])
    eps = np.array([0.1, -0.2, 0.3, 0.02, 0.5, -0.4])
    out = oracle.se3_left_update(eps, T.reshape(-1)).reshape(3, 4)
    Rw = sl.so3_exp(eps[3:])
    assert np.allclose(out[:, :3], Rw @ T[:, :3], atol=1e-14) and np.allclose(out[:, 3], Rw @ T[:, 3] + eps[:3], atol=1e-14)
    tiny = oracle.se3_left_update(np.array([0, 0, 0, 1e-10, -2e-10, 3e-10]), T.reshape(-1)).reshape(3, 4)
    assert np.allclose(tiny, T, atol=1e-9)


def test_noise_free_scene_converges_to_ground_truth(oracle):
    win, gt = small(4, px_noise=0.0, edge_noise=(0.08, 0.015), lm_noise=0.08)
    # observations are float32 pixels: the attainable residual floor is ~1e-5 px
    prm = sl.make_params(max_iters=40, robust=False)
    out = oracle.lm_solve(win, prm)
    r = out["res"]
    assert r.err_final < 1e-6 * r.err_init and np.sqrt(r.err_final / r.n_obs) < 1e-4
    unk = win.edge_unknown_slot >= 0
    assert np.abs(out["poses"][unk] - gt["gt_pose"][unk]).max() < 1e-4
    assert np.abs(out["lms"] - gt["gt_lm"]).max() < 2e-3
    assert np.array_equal(out["poses"][~unk], win.edge_inv_pose[~unk])     # fixed edges untouched


def test_error_is_monotone_and_lambda_schedule(oracle):
    win, gt = small(5)
    prm = sl.make_params()
    out = oracle.lm_solve(win, prm)
    tr = out["trace"]
    assert out["res"].accepted >= 2 and len(tr) == out["res"].iters
    err = out["res"].err_init
    lam = tr[0].lambda_
    for t in tr:
        assert t.err_before == err and np.isclose(t.lambda_, lam, rtol=1e-12)
        if t.accepted:
            assert t.err_after < t.err_before
            err, lam = t.err_after, lam * 0.1
            nu = 2.0
        else:
            lam, nu = (lam * 2.0, 4.0) if "nu" not in dir() else (lam * nu, nu * 2)
    assert np.isclose(out["res"].err_final, err)
    # with sigma=0.5 px noise on 4 components the robustified error per observation settles near 4*0.25
    assert 0.5 < out["res"].err_final / out["res"].n_obs < 1.5


def test_stage1_edges_only_and_nothing_to_do(oracle):
    """stage-1 shape (optimize_edges on the new edge only, landmarks fixed) and the empty problem."""
    win, gt = small(6)
    w1 = sl.Window(np.where(win.edge_unknown_slot == win.E - 1, 0, -1), win.edge_inv_pose,
                   -np.ones(len(win.lm_pos), np.int32), gt["gt_lm"], win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    assert w1.E == 1 and w1.L == 0
    out = oracle.lm_solve(w1, sl.make_params())
    assert out["res"].accepted >= 1 and out["res"].err_final < out["res"].err_init
    assert np.array_equal(out["lms"], gt["gt_lm"])
    w0 = sl.Window(-np.ones(len(win.edge_unknown_slot), np.int32), win.edge_inv_pose, -np.ones(len(win.lm_pos), np.int32),
                   win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    out = oracle.lm_solve(w0, sl.make_params())
    assert out["res"].stop_reason == 3 and out["res"].iters == 0 and out["res"].err_final == out["res"].err_init


def test_golden_lm_trace(oracle):
    g = np.load(os.path.join(GOLD, "lm_case0.npz"))
    win = sl.Window(g["edge_unknown_slot"], g["edge_inv_pose"], g["lm_unknown_slot"], g["lm_pos"], g["obs_lm"], g["obs_px"],
                    g["obs_path_off"], g["path_edge"])
    out = oracle.lm_solve(win, sl.make_params())
    assert out["res"].iters == int(g["iters"]) and out["res"].accepted == int(g["accepted"])
    assert np.allclose([t.err_after for t in out["trace"]], g["err_after"], rtol=1e-9)
    assert np.allclose(out["poses"], g["final_poses"], rtol=1e-9, atol=1e-11)
    assert np.allclose(out["lms"], g["final_lms"], rtol=1e-9, atol=1e-11)
