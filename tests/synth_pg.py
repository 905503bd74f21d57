"""Synthetic pose graphs for the global pose-graph optimisation tests: a keyframe trajectory that drives a closed loop,
odometry edges with drift, loop-closure edges; poses row-major [R|t]."""
import numpy as np


def rotvec(w):
    w = np.asarray(w, float)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def pose(R, t):
    return np.hstack([R, np.asarray(t, float).reshape(3, 1)]).reshape(12)


def mul(a, b):
    A, B = a.reshape(3, 4), b.reshape(3, 4)
    return pose(A[:, :3] @ B[:, :3], A[:, :3] @ B[:, 3] + A[:, 3])


def inv(a):
    A = a.reshape(3, 4)
    return pose(A[:, :3].T, -A[:, :3].T @ A[:, 3])


def make_loop(seed, n=60, radius=20.0, noise_t=0.05, noise_r=0.004, n_closures=3, edge_forward=True):
    """Ground-truth poses on a circle (heading along the tangent, small roll/pitch wobble), odometry edges k -> k+1 measured
    with noise, `n_closures` exact loop-closure edges between the end and the start.  Returns dict(gt, init, edge_ij, edge_z):
    init = the noisy odometry chained from node 0 (what the spanning tree of the relative map gives)."""
    rng = np.random.default_rng(seed)
    gt = []
    for k in range(n):
        a = 2 * np.pi * k / n
        R = rotvec([0, 0, a + np.pi / 2]) @ rotvec(0.05 * np.sin(3 * a) * np.array([1.0, 0, 0]))
        gt.append(pose(R, [radius * np.cos(a), radius * np.sin(a), 0.3 * np.sin(2 * a)]))
    gt0 = inv(gt[0])
    gt = np.array([mul(gt0, g) for g in gt])                       # root at the identity
    ij, z = [], []
    for k in range(n - 1):
        rel = mul(inv(gt[k]), gt[k + 1])
        meas = mul(rel, pose(rotvec(rng.normal(0, noise_r, 3)), rng.normal(0, noise_t, 3)))
        if edge_forward or k % 2 == 0:
            ij.append((k, k + 1)); z.append(meas)
        else:
            ij.append((k + 1, k)); z.append(inv(meas))               # some edges stored the other way round, as k2k edges are
    init = [gt[0]]
    for k in range(n - 1):
        i, j = ij[k]
        init.append(mul(init[-1], z[k] if (i, j) == (k, k + 1) else inv(z[k])))
    for c in range(n_closures):
        i, j = n - 1 - c, c
        ij.append((i, j)); z.append(mul(inv(gt[i]), gt[j]))
    return dict(gt=gt, init=np.array(init), edge_ij=np.array(ij, np.int32), edge_z=np.array(z))
