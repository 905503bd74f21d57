"""Synthetic tracked stereo pairs for the change-in-pose filter: 3-D points seen by the OTHER keyframe, a known motion to
THIS keyframe, pixel noise, gross outliers, and a status vector with some matches already rejected by earlier filters."""
import numpy as np

FX, CX, CY, B = 707.0912, 601.8873, 183.1104, 0.54


def project(P):
    return np.stack([CX + FX * P[:, 0] / P[:, 2], CY + FX * P[:, 1] / P[:, 2],
                     CX + FX * (P[:, 0] - B) / P[:, 2], CY + FX * P[:, 1] / P[:, 2]], 1)


def rodrigues(w):
    th = float(np.linalg.norm(w))
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def make_case(seed, n_this=360, n_other=400, n_outliers=25, noise=0.3, w=(0.01, -0.02, 0.005), t=(0.05, -0.02, -1.0),
              frac_pre_rejected=0.15):
    """Returns dict(this_obs, other_obs, train_idx, status, R, t, outlier mask)."""
    rng = np.random.default_rng(0xC1B0000 + seed)
    P = np.stack([rng.uniform(-9, 9, n_other), rng.uniform(-2.5, 2.0, n_other), rng.uniform(5, 45, n_other)], 1)
    other_obs = project(P)
    other_obs[:, 3] = other_obs[:, 1]
    other_obs = np.round(other_obs).astype(np.float32)          # keypoints sit on integer pixels
    train_idx = rng.permutation(n_other)[:n_this].astype(np.int32)
    R, t = rodrigues(np.asarray(w, float)), np.asarray(t, float)
    Q = P[train_idx] @ R.T + t
    this_obs = project(Q) + rng.normal(0, noise, (n_this, 4))
    out = np.zeros(n_this, bool)
    out[rng.permutation(n_this)[:n_outliers]] = True
    this_obs[out] += rng.uniform(6, 40, (int(out.sum()), 4)) * rng.choice([-1, 1], (int(out.sum()), 4))
    status = np.zeros(n_this, np.int32)
    pre = rng.random(n_this) < frac_pre_rejected
    status[pre] = rng.integers(1, 4, int(pre.sum()))            # NON_TRACKED / REJ_SLOPE / REJ_ORB from the earlier filters
    return dict(this_obs=this_obs.astype(np.float32), other_obs=other_obs, train_idx=train_idx, status=status, R=R, t=t, outlier=out)


def identity_pose():
    return np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(12)
