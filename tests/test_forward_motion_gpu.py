"""GPU: the estimator on FORWARD motion (KITTI-like geometry: a camera driving through a textured corridor, rendered by ray
casting) -- radial optical flow, scale change, a non-degenerate epipolar geometry for the fundamental-matrix stage, and a
gentle turn: the odometry stand-in, the keyframe policy and the relative BA must recover the driven path."""
import os

import numpy as np
import pytest

import bow_vocab as bv
import synth

pytestmark = pytest.mark.gpu


def run(ctx, tmp_path, frames, **options):
    from ssk_amd import capi
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    voc = os.path.join(tmp_path, "voc.yml.gz")
    bv.write_yaml(bv.train([ctx.frontend_run(l, r, prm)["desc_left"] for l, r in frames[::3]], k=8, L=3, seed=5), voc)
    slam = capi.Slam(ctx, synth.W, synth.H, "kitti", **options)
    slam.load_vocabulary(voc)
    reps = [slam.process_frame(l, r) for l, r in frames]
    X, info = slam.optimize_pose_graph()
    counts = slam.counts()
    slam.close()
    return reps, X, counts


@pytest.mark.parametrize("seed", [2, 5, 8])
def test_straight_drive(pkg, ctx, tmp_path, seed):
    n, step = 13, 0.5
    frames, poses = synth.synth_forward_sequence(seed, n, step=step)
    reps, X, (nk, ne, nl) = run(ctx, tmp_path, frames)
    for k, r in enumerate(reps[1:], 1):
        T = np.array(r.vo_pose[:]).reshape(3, 4)
        assert r.vo_valid == 1 and r.tracked_from_last_frame >= 60, (k, r.tracked_from_last_frame)
        assert abs(T[2, 3] - step) < 0.05 and abs(T[0, 3]) < 0.05 and abs(T[1, 3]) < 0.05, (k, T[:, 3])   # forward = camera +z
        assert np.abs(T[:, :3] - np.eye(3)).max() < 5e-3
    kf_frames = [k for k, r in enumerate(reps) if r.inserted_kf]
    assert kf_frames[0] == 0 and 3 <= nk <= 5 and ne == nk - 1, kf_frames            # a keyframe every 2 m (KITTI limit) or so
    assert all(r.highest_tracked >= 15 for r in reps[1:] if r.inserted_kf)              # (every new keyframe shares landmarks with an old one)
    # keyframe poses in the robot frame of keyframe 0: x forward (camera_pose_on_robot of the ini), y left, z up
    for i, k in enumerate(kf_frames):
        p = X[i].reshape(3, 4)
        assert abs(p[0, 3] - step * k) < 0.03 * step * k + 0.05 and abs(p[1, 3]) < 0.08 and abs(p[2, 3]) < 0.08, (k, p[:, 3])
    for r in reps:
        if r.inserted_kf and r.kf_id > 0:
            assert r.total_sqr_error_final <= r.total_sqr_error_init


def test_keyframe_without_tracked_features_keeps_the_odometry_increment(pkg, ctx, tmp_path):
    """Two of the reference's quirks meeting (both kept): fewer than 15 survivors in front of the fundamental-matrix or the pose filter
    make the whole keyframe pair invalid -- ZERO tracked features (EST.cpp:1591-1595, 1619-1623, 1647) -- and the keyframe is inserted all
    the same, its new edge starting from the odometry increment of the CAMERA frame handed over as a ROBOT-frame pose (EST.cpp:772-778).
    With landmarks in common the LM pulls the edge where it belongs (test_straight_drive); with none it stays: on this sequence (seed 1:
    the second keyframe pair ends just under the gate) the 2 m driven forward (robot x) show up as 2 m upwards (robot z = camera -y... the
    camera's forward axis).  A test of what the estimator does, not of what one would like: over 16 rendered drives this is the only one
    that meets the gate."""
    n, step = 13, 0.5
    frames, poses = synth.synth_forward_sequence(1, n, step=step)
    reps, X, (nk, ne, nl) = run(ctx, tmp_path, frames)
    kfs = [(k, r) for k, r in enumerate(reps) if r.inserted_kf]
    for i, (k, r) in enumerate(kfs):
        p = X[i].reshape(3, 4)
        ok = abs(p[0, 3] - step * k) < 0.03 * step * k + 0.05 and abs(p[1, 3]) < 0.08 and abs(p[2, 3]) < 0.08
        if i == 0 or all(q.highest_tracked >= 15 for _, q in kfs[1:i + 1]):
            assert ok, (k, p[:, 3])                    # constrained all the way from keyframe 0
        elif r.highest_tracked == 0:
            assert not ok and np.linalg.norm(p[:, 3] - [step * k, 0, 0]) > 1.0, (k, p[:, 3])


def test_gentle_turn(pkg, ctx, tmp_path):
    """0.8 degrees of yaw per frame while driving: the heading of the last keyframe follows."""
    n, step, yaw = 13, 0.5, np.deg2rad(0.8)
    frames, poses = synth.synth_forward_sequence(2, n, step=step, yaw_per_frame=yaw)
    reps, X, (nk, ne, nl) = run(ctx, tmp_path, frames)
    assert all(r.vo_valid == 1 for r in reps[1:])
    kf_frames = [k for k, r in enumerate(reps) if r.inserted_kf]
    k = kf_frames[-1]
    R_gt, c_gt = poses[k]
    # camera frame (x right, y down, z forward) -> robot frame (x forward, y left, z up): p_robot = S p_cam
    S = np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
    p = X[-1].reshape(3, 4)
    assert np.linalg.norm(p[:, 3] - S @ c_gt) < 0.04 * np.linalg.norm(c_gt) + 0.05, (p[:, 3], S @ c_gt)
    assert np.abs(p[:, :3] - S @ R_gt @ S.T).max() < 0.01
