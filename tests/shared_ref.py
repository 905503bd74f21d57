"""numpy restatement of the shared-landmark exchange (BASELINE configs[4]): what one submap contributes for a landmark
it observes but does not own -- g = sum Jf^T w r and H = sum Jf^T Jf over its observations at its current state, with the
observation model and the IRLS weight of the LM contract (oracle/oracle_lm.c L1-L4) -- and the consensus update every rank
applies to the gathered blocks: x <- x + (sum_r H_r + lambda I)^-1 sum_r g_r, summed in rank order."""
import numpy as np


def _cam(prm):
    S = np.array(prm.sensor_pose[:]).reshape(3, 4)
    return S[:, :3], S[:, 3]


def blocks(win, prm, poses, lms, shared_idx):
    """(n_shared, 12): position (3), g (3), upper H (6) of every shared landmark from this window's observations."""
    Rs, ts = _cam(prm)
    out = np.zeros((len(shared_idx), 12))
    where = {int(li): s for s, li in enumerate(shared_idx) if li >= 0}
    for s, li in enumerate(shared_idx):
        if li >= 0:
            out[s, :3] = lms[li]
    k2 = prm.kernel_param ** 2
    for o in range(len(win.obs_lm)):
        li = int(win.obs_lm[o])
        if li not in where:
            continue
        q = lms[li].copy()
        R = np.eye(3)
        for c in win.path_edge[win.obs_path_off[o]:win.obs_path_off[o + 1]][::-1]:
            T = poses[c >> 1].reshape(3, 4)
            if c & 1:
                q = T[:, :3].T @ (q - T[:, 3]); R = T[:, :3].T @ R
            else:
                q = T[:, :3] @ q + T[:, 3]; R = T[:, :3] @ R
        pc = Rs.T @ (q - ts)
        iz = 1.0 / pc[2]
        h = np.array([prm.cx_l + prm.fx_l * pc[0] * iz, prm.cy_l + prm.fy_l * pc[1] * iz,
                      prm.cx_r + prm.fx_r * (pc[0] - prm.baseline) * iz, prm.cy_r + prm.fy_r * pc[1] * iz])
        r = win.obs_px[o].astype(np.float64) - h
        s2 = float(r @ r)
        w = (1.0 + s2 / k2) ** -0.25 if prm.use_robust_kernel else 1.0
        D = np.array([[prm.fx_l * iz, 0, -prm.fx_l * pc[0] * iz * iz], [0, prm.fy_l * iz, -prm.fy_l * pc[1] * iz * iz],
                      [prm.fx_r * iz, 0, -prm.fx_r * (pc[0] - prm.baseline) * iz * iz], [0, prm.fy_r * iz, -prm.fy_r * pc[1] * iz * iz]])
        Jf = w * D @ Rs.T @ R
        Hm = Jf.T @ Jf
        s = where[li]
        out[s, 3:6] += Jf.T @ (w * r)
        out[s, 6:12] += [Hm[0, 0], Hm[0, 1], Hm[0, 2], Hm[1, 1], Hm[1, 2], Hm[2, 2]]
    return out


def consensus(gathered, positions, lam=0.0):
    """gathered (n_ranks, n_shared, 12) in rank order, positions (n_shared, 3) -> updated positions."""
    out = positions.copy()
    for s in range(gathered.shape[1]):
        g = np.zeros(3); H = np.zeros((3, 3))
        for rk in range(gathered.shape[0]):
            b = gathered[rk, s]
            g += b[3:6]
            H += np.array([[b[6], b[7], b[8]], [b[7], b[9], b[10]], [b[8], b[10], b[11]]])
        if H[0, 0] + H[1, 1] + H[2, 2] > 0:
            out[s] = positions[s] + np.linalg.solve(H + lam * np.eye(3), g)
    return out
