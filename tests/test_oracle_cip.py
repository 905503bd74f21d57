"""CPU: the oracle's change-in-pose filter (oracle/oracle_cip.c, contract P1-P6) against known answers: it recovers a
known motion, flags the gross outliers and nothing else, leaves already rejected matches alone, and follows the literal
in-tree rules of EST.cpp:2113-2177 (invalid estimate => every tracked match rejected)."""
import os

import numpy as np

import oracle_lib as ol
import synth_cip as sc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_recovers_motion_and_flags_outliers(oracle):
    c = sc.make_case(1)
    st, res, out = oracle.change_in_pose(c["this_obs"], c["other_obs"], c["train_idx"], c["status"], sc.identity_pose(), ol.kitti_cip_params())
    T = np.array(out.pose[:]).reshape(3, 4)
    assert out.valid == 1 and out.iters_initial >= 2 and out.iters_final >= 1
    assert np.abs(T[:, :3] - c["R"]).max() < 2e-3 and np.abs(T[:, 3] - c["t"]).max() < 3e-2
    trk = c["status"] == 0
    assert out.n_pairs == int(trk.sum()) == len(res)
    assert np.array_equal(st[~trk], c["status"][~trk])                       # earlier verdicts are never touched
    assert np.all(st[trk & c["outlier"]] == 5)                              # sREJ_CHANGE_POSE
    assert (st[trk & ~c["outlier"]] == 5).mean() < 0.02                      # inliers stay (noise 0.3 px vs gate 30 px^2)
    assert np.array_equal(st[trk] == 5, res > 30.0) and out.n_rejected == int((res > 30.0).sum())
    assert out.n_inliers == int((res <= 15.0).sum()) or abs(out.n_inliers - int((res <= 15.0).sum())) <= 2


def test_noise_free_is_exact_and_converges_fast(oracle):
    c = sc.make_case(2, noise=0.0, n_outliers=0)
    # un-round the other keyframe's pixels so that the 3-D points are exact
    st, res, out = oracle.change_in_pose(c["this_obs"], c["other_obs"], c["train_idx"], c["status"], sc.identity_pose(),
                                         ol.kitti_cip_params(use_robust_kernel=0))
    assert out.valid == 1 and out.n_rejected <= 2 and out.iters_initial <= 10
    assert np.median(res) < 2.0                                               # only the integer-pixel rounding of the other keyframe remains


def test_invalid_estimate_rejects_every_tracked_match(oracle):
    c = sc.make_case(3, n_this=12, n_other=20, n_outliers=0)
    st0 = np.ones(12, np.int32)
    st0[:4] = 0                                                               # 4 tracked pairs < bad_tracking_th = 5
    st, res, out = oracle.change_in_pose(c["this_obs"], c["other_obs"], c["train_idx"], st0, sc.identity_pose(), ol.kitti_cip_params())
    assert out.valid == 0 and np.all(st[:4] == 5) and np.all(st[4:] == 1) and out.n_rejected == 4


def test_good_initial_pose_needs_fewer_iterations(oracle):
    c = sc.make_case(4)
    T0 = np.hstack([c["R"], c["t"][:, None]]).reshape(12)
    _, _, cold = oracle.change_in_pose(c["this_obs"], c["other_obs"], c["train_idx"], c["status"], sc.identity_pose(), ol.kitti_cip_params())
    _, _, warm = oracle.change_in_pose(c["this_obs"], c["other_obs"], c["train_idx"], c["status"], T0, ol.kitti_cip_params())
    assert warm.valid and cold.valid and warm.iters_initial <= cold.iters_initial
    assert np.abs(np.array(warm.pose[:]) - np.array(cold.pose[:])).max() < 1e-3


def test_golden_cip(oracle):
    """committed fixture (tests/golden/make_golden.py): guards the oracle against regressions; not a reference output."""
    g = np.load(os.path.join(GOLD, "cip_case0.npz"))
    st, res, out = oracle.change_in_pose(g["this_obs"], g["other_obs"], g["train_idx"], g["status"], sc.identity_pose(), ol.kitti_cip_params())
    assert np.array_equal(st, g["exp_status"])
    assert [out.iters_initial, out.iters_final, out.n_inliers, out.n_rejected] == list(g["exp_iters"])
    assert np.allclose(res, g["exp_residuals"], rtol=1e-9, atol=1e-12) and np.allclose(np.array(out.pose[:]), g["exp_pose"], rtol=0, atol=1e-12)
