"""Seeded synthetic relative-BA problems shaped like BASELINE.json configs[2] ("200-KF / 5k-landmark SRBA
submap, local LM window"): keyframes on a smooth trajectory, fixed-size submaps joined star-of-stars as
mySRBA::edge_creation_policy builds them (reference src/srba-stereo-slam.h:144-308), landmarks relative to
the first keyframe that sees them, stereo observations with N(0, 0.5^2) pixel noise (EST.cpp:1157).

Conventions are those of include/ssk_capi.h (ssk_lm_window): edge inv_pose = pose of `from` seen from `to`.
"""
import ctypes as C
from collections import deque

import numpy as np

KITTI_CAM = dict(fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104, baseline=0.54, w=1226, h=370)
# camera_pose_on_robot (+) (0,0,0,-90deg,0,-90deg)  (EST.cpp:1105-1112): camera z forward, x right, y down
R_SENSOR = np.array([[0.0, 0, 1], [-1, 0, 0], [0, -1, 0]])


def se3(R, t):
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    return T


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-8:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


class LmWindowC(C.Structure):
    _fields_ = [("n_edges_total", C.c_int32), ("edge_unknown_slot", C.c_void_p), ("edge_inv_pose", C.c_void_p),
                ("n_lms_total", C.c_int32), ("lm_unknown_slot", C.c_void_p), ("lm_pos", C.c_void_p),
                ("n_obs", C.c_int32), ("obs_lm", C.c_void_p), ("obs_px", C.c_void_p),
                ("obs_path_off", C.c_void_p), ("path_edge", C.c_void_p)]


class LmParamsC(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx_l", "fy_l", "cx_l", "cy_l", "fx_r", "fy_r", "cx_r", "cy_r", "baseline")] + \
               [("sensor_pose", C.c_double * 12), ("max_iters", C.c_int32), ("use_robust_kernel", C.c_int32),
                ("kernel_param", C.c_double), ("tau", C.c_double), ("max_lambda", C.c_double),
                ("max_error_per_obs_to_stop", C.c_double), ("converged_rel_decrease", C.c_double)]


class LmResultC(C.Structure):
    _fields_ = [("iters", C.c_int32), ("accepted", C.c_int32), ("stop_reason", C.c_int32), ("n_obs", C.c_int32),
                ("err_init", C.c_double), ("err_final", C.c_double), ("lambda_final", C.c_double)]


class LmTraceC(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("err_before", C.c_double), ("err_after", C.c_double),
                ("delta_p_norm", C.c_double), ("delta_f_norm", C.c_double), ("accepted", C.c_int32),
                ("cholesky_ok", C.c_int32)]


def make_params(max_iters=20, robust=True, kernel=1.5, sensor_t=(0.0, 0.0, 0.0), conv=1e-9):
    k = KITTI_CAM
    S = np.hstack([R_SENSOR, np.array(sensor_t, float).reshape(3, 1)])
    return LmParamsC(k["fx"], k["fy"], k["cx"], k["cy"], k["fx"], k["fy"], k["cx"], k["cy"], k["baseline"],
                     (C.c_double * 12)(*S.reshape(-1)), max_iters, int(robust), kernel, 1e-3, 1e20, 1e-9, conv)


class Window:
    """numpy arrays of one ssk_lm_window + its C view (arrays kept alive by the object)."""

    def __init__(self, edge_unknown_slot, edge_inv_pose, lm_unknown_slot, lm_pos, obs_lm, obs_px, obs_path_off, path_edge):
        self.edge_unknown_slot = np.ascontiguousarray(edge_unknown_slot, np.int32)
        self.edge_inv_pose = np.ascontiguousarray(edge_inv_pose, np.float64).reshape(-1, 12)
        self.lm_unknown_slot = np.ascontiguousarray(lm_unknown_slot, np.int32)
        self.lm_pos = np.ascontiguousarray(lm_pos, np.float64).reshape(-1, 3)
        self.obs_lm = np.ascontiguousarray(obs_lm, np.int32)
        self.obs_px = np.ascontiguousarray(obs_px, np.float32).reshape(-1, 4)
        self.obs_path_off = np.ascontiguousarray(obs_path_off, np.int32)
        self.path_edge = np.ascontiguousarray(path_edge, np.int32)
        self.E = int((self.edge_unknown_slot >= 0).sum())
        self.L = int((self.lm_unknown_slot >= 0).sum())

    def c(self):
        p = lambda a: a.ctypes.data if a.size else 0
        return LmWindowC(len(self.edge_unknown_slot), p(self.edge_unknown_slot), p(self.edge_inv_pose),
                         len(self.lm_unknown_slot), p(self.lm_unknown_slot), p(self.lm_pos),
                         len(self.obs_lm), p(self.obs_lm), p(self.obs_px), p(self.obs_path_off), p(self.path_edge))


def make_submap_problem(seed, n_kf=30, n_lm=800, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250,
                        px_noise=0.5, edge_noise=(0.05, 0.01), lm_noise=0.05, root=None, lm_fwd=(4.0, 45.0), z_vis=(3.0, 60.0), shared=None):
    """Returns (Window with a perturbed initial state, dict with ground truth).
    shared (BASELINE configs[4]): dict(points (n, 3) in the frame of a common ANCHOR keyframe, init (n, 3) their initial estimates,
    G 4x4 = the anchor seen from this submap's root keyframe, use (n,) bool: which of them this submap observes).  The anchor hangs on
    the root by a FIXED edge at its true value; the landmarks are based at the anchor, observed from the keyframes of the local area
    that see them, and come last in the window's landmark list, NOT optimised (lm_unknown_slot -1): gt["shared_idx"][s] = index of
    shared landmark s in this window, or -1."""
    rng = np.random.default_rng(0xBA5E0000 + seed)
    k = KITTI_CAM
    # ---- trajectory: ~1 m per keyframe, gentle yaw; robot x forward, z up ----
    Tw = []
    x, y, yaw = 0.0, 0.0, 0.0
    for i in range(n_kf):
        Tw.append(se3(rot_z(yaw), [x, y, 0.0]))
        yaw += rng.normal(0.0, 0.02) + 0.01
        x += np.cos(yaw) * 1.0
        y += np.sin(yaw) * 1.0
    # ---- graph: fixed-size submaps, star-of-stars (SRBA.h:176-306) ----
    edges = []          # (from, to)
    center = 0
    members = 0
    centers = [0]
    adj = {i: [] for i in range(n_kf)}
    for i in range(1, n_kf):
        if members < submap_size - 1:
            edges.append((center, i))
            members += 1
        else:
            edges.append((center, i))       # new centre linked to the previous centre
            center, members = i, 0
            centers.append(i)
        a, b = edges[-1]
        adj[a].append((b, len(edges) - 1))
        adj[b].append((a, len(edges) - 1))
    inv_pose_gt = [np.linalg.inv(Tw[b]) @ Tw[a] for a, b in edges]      # `from` seen from `to`
    # ---- landmarks along the corridor, observed by keyframes that see them in both cameras ----
    S = se3(R_SENSOR, [0, 0, 0])
    # scattered around the trajectory: each landmark sits ahead of a random keyframe (robot frame: x fwd, y left, z up)
    anchor = rng.integers(0, n_kf, n_lm)
    local = np.column_stack([rng.uniform(lm_fwd[0], lm_fwd[1], n_lm), rng.uniform(-15, 15, n_lm), rng.uniform(-2, 6, n_lm), np.ones(n_lm)])
    lm_world = np.array([(Tw[a] @ p)[:3] for a, p in zip(anchor, local)])
    base_of = -np.ones(n_lm, int)
    obs = []            # (kf, lm, px4)

    lm_h = np.column_stack([lm_world, np.ones(n_lm)])

    def project_all(T_kf):
        pc = (np.linalg.inv(T_kf @ S) @ lm_h.T)[:3].T
        z = np.where(pc[:, 2] > 1e-9, pc[:, 2], 1.0)
        u = k["cx"] + k["fx"] * pc[:, 0] / z
        v = k["cy"] + k["fy"] * pc[:, 1] / z
        ur = k["cx"] + k["fx"] * (pc[:, 0] - k["baseline"]) / z
        ok = (pc[:, 2] >= z_vis[0]) & (pc[:, 2] <= z_vis[1]) & (u >= 0) & (u < k["w"]) & (ur >= 0) & (ur < k["w"]) & (v >= 0) & (v < k["h"])
        return ok, np.column_stack([u, v, ur, v])

    for i in range(n_kf):
        ok, px = project_all(Tw[i])
        vis = [(int(l), px[l]) for l in np.flatnonzero(ok)]
        if len(vis) > obs_per_kf:
            sel = rng.choice(len(vis), obs_per_kf, replace=False)
            vis = [vis[s] for s in sorted(sel)]
        for l, p in vis:
            if base_of[l] < 0:
                base_of[l] = i
            obs.append((i, l, p + rng.normal(0, px_noise, 4) if px_noise > 0 else p))
    # ---- paths by BFS (tree, so unique), dropped beyond max_tree_depth ----
    def path(src, dst):
        prev = {src: None}
        dq = deque([src])
        while dq:
            u = dq.popleft()
            if u == dst:
                break
            for v, e in adj[u]:
                if v not in prev:
                    prev[v] = (u, e)
                    dq.append(v)
        out, u = [], dst
        while prev[u] is not None:
            pu, e = prev[u]
            out.append((e, pu, u))
            u = pu
        return out[::-1]     # steps from src towards dst: (edge, cur, next)

    dist_from_root = {}
    root = n_kf - 1 if root is None else root
    dq = deque([(root, 0)])
    dist_from_root[root] = 0
    while dq:
        u, d = dq.popleft()
        for v, e in adj[u]:
            if v not in dist_from_root:
                dist_from_root[v] = d + 1
                dq.append((v, d + 1))
    area = {kf for kf, d in dist_from_root.items() if d <= opt_depth}
    opt_edges = [ei for ei, (a, b) in enumerate(edges) if a in area and b in area]
    win_obs = []
    for kf, l, p in obs:
        if kf not in area:          # observed from the local area; the base keyframe may lie outside it
            continue                # (then the path runs over fixed edges, as in SRBA's optimize_local_area)
        steps = path(kf, base_of[l])
        if len(steps) > max_tree_depth:
            continue
        win_obs.append((kf, l, p, steps))
    # ---- shared landmarks: based at the anchor keyframe n_kf, reached over the fixed edge (anchor -> root) ----
    shared_lm0 = n_lm
    if shared is not None:
        G = np.asarray(shared["G"], float)
        anchor_edge = len(edges)
        edges.append((n_kf, root))
        inv_pose_gt.append(G)                                  # pose of `from` (the anchor) seen from `to` (the root)
        pts = np.asarray(shared["points"], float)
        for s_i in np.flatnonzero(shared["use"]):
            pw = Tw[root] @ G @ np.append(pts[s_i], 1.0)
            n_seen = 0
            for kf in sorted(area):
                pc = (np.linalg.inv(Tw[kf] @ S) @ pw)[:3]
                if not (z_vis[0] <= pc[2] <= z_vis[1]):
                    continue
                u, v, ur = k["cx"] + k["fx"] * pc[0] / pc[2], k["cy"] + k["fy"] * pc[1] / pc[2], k["cx"] + k["fx"] * (pc[0] - k["baseline"]) / pc[2]
                if not (0 <= u < k["w"] and 0 <= ur < k["w"] and 0 <= v < k["h"]):
                    continue
                steps = path(kf, root) + [(anchor_edge, root, n_kf)]
                if len(steps) > max_tree_depth:
                    continue
                px = np.array([u, v, ur, v])
                win_obs.append((kf, shared_lm0 + int(s_i), px + rng.normal(0, px_noise, 4) if px_noise > 0 else px, steps))
                n_seen += 1
    used_edges = sorted({e for _, _, _, st in win_obs for e, _, _ in st} | set(opt_edges))
    eidx = {e: i for i, e in enumerate(used_edges)}
    used_lms = sorted({l for _, l, _, _ in win_obs})
    lidx = {l: i for i, l in enumerate(used_lms)}
    edge_slot = -np.ones(len(used_edges), np.int32)
    for s, e in enumerate(opt_edges):
        edge_slot[eidx[e]] = s
    lm_slot = np.arange(len(used_lms), dtype=np.int32)       # every landmark of the area is optimised
    if shared is not None:
        lm_slot[[i for i, l in enumerate(used_lms) if l >= shared_lm0]] = -1      # ... but the shared ones: moved by the exchange, not by this window
        lm_slot[lm_slot >= 0] = np.arange(int((lm_slot >= 0).sum()))
    gt_pose = np.array([inv_pose_gt[e][:3, :].reshape(-1) for e in used_edges])
    gt_lm = np.array([(np.linalg.inv(Tw[base_of[l]]) @ np.append(lm_world[l], 1.0))[:3] if l < shared_lm0 else np.asarray(shared["points"], float)[l - shared_lm0]
                      for l in used_lms]).reshape(-1, 3)
    obs_lm = np.array([lidx[l] for _, l, _, _ in win_obs], np.int32)
    obs_px = np.array([p for _, _, p, _ in win_obs], np.float32).reshape(-1, 4)
    off, pe = [0], []
    for _, _, _, st in win_obs:
        for e, cur, nxt in st:
            a, b = edges[e]
            # inv_pose maps `from`-frame points to the `to` frame: applying it goes next=from -> cur=to
            pe.append(eidx[e] * 2 + (0 if (nxt == a and cur == b) else 1))
        off.append(len(pe))
    # ---- perturbed initial state ----
    init_pose = gt_pose.copy()
    for i in range(len(used_edges)):
        if edge_slot[i] >= 0:
            T = se3(gt_pose[i].reshape(3, 4)[:, :3], gt_pose[i].reshape(3, 4)[:, 3])
            dT = se3(so3_exp(rng.normal(0, edge_noise[1], 3)), rng.normal(0, edge_noise[0], 3))
            init_pose[i] = (dT @ T)[:3, :].reshape(-1)
    init_lm = gt_lm + rng.normal(0, lm_noise, gt_lm.shape) * (1 + np.abs(gt_lm[:, :1]) / 10.0)
    out = dict(gt_pose=gt_pose, gt_lm=gt_lm, n_kf=n_kf, edges=[edges[e] for e in used_edges], area=sorted(area))
    if shared is not None:
        idx = -np.ones(len(shared["points"]), np.int32)
        for i, l in enumerate(used_lms):
            if l >= shared_lm0:
                idx[l - shared_lm0] = i
                init_lm[i] = np.asarray(shared["init"], float)[l - shared_lm0]      # every submap starts from the same estimate
        out["shared_idx"] = idx
    win = Window(edge_slot, init_pose, lm_slot, init_lm, obs_lm, obs_px, off, pe)
    return win, out


def make_shared_submaps(n_sub=8, n_shared=256, seed0=50, per_landmark=2, only=None, **kw):
    """BASELINE configs[4]: n_sub independent config-3-shaped submaps (seeds seed0 ..) + n_shared landmarks of ONE identity each,
    observed from `per_landmark` >= 2 of the submaps (landmark s from submaps s, s+1, .. mod n_sub): the submaps' newest keyframes stand
    around a common anchor keyframe (known, fixed relative poses) and the shared landmarks lie in front of it.
    only: build just these submaps (every rank of a multi-GPU run builds its own; the shared set is the same for all).
    Returns ([Window], [shared_idx arrays], dict(points, init))."""
    rng = np.random.default_rng(0x5AAE0000 + seed0)
    x = rng.uniform(5.0, 14.0, n_shared)
    pts = np.column_stack([x, rng.uniform(-0.35, 0.35, n_shared) * x, rng.uniform(-0.12, 0.2, n_shared) * x])
    init = pts + rng.normal(0, 0.05, pts.shape) * (1 + x[:, None] / 10.0)
    wins, idxs = [], []
    for r in range(n_sub):
        use = np.array([(r - s) % n_sub < per_landmark for s in range(n_shared)])
        G = se3(rot_z(rng.normal(0, 0.05)), [rng.normal(0, 0.3), rng.normal(0, 0.3), 0.0])      # the anchor seen from this submap's root
        if only is not None and r not in only:
            continue
        w, gt = make_submap_problem(seed0 + r, shared=dict(points=pts, init=init, G=G, use=use), **kw)
        wins.append(w); idxs.append(gt["shared_idx"])
    return wins, idxs, dict(points=pts, init=init)


def join_windows(wins, idxs):
    """The union of the submaps as ONE window: edges and private landmarks side by side, every shared landmark once and OPTIMISED --
    what a single process would solve.  Returns (Window, per-submap edge offsets, per-submap landmark index maps, joint index of
    every shared landmark)."""
    n_shared = len(idxs[0])
    e_off, slots, poses, lm_map, lm_pos, lm_slot = [], [], [], [], [], []
    ne = 0
    E = 0
    shared_joint = -np.ones(n_shared, np.int64)
    for w, idx in zip(wins, idxs):
        e_off.append(ne)
        sl = w.edge_unknown_slot.copy()
        sl[sl >= 0] += E
        E += w.E
        slots.append(sl); poses.append(w.edge_inv_pose); ne += len(sl)
        inv = {int(li): s for s, li in enumerate(idx) if li >= 0}
        m = np.zeros(len(w.lm_pos), np.int64)
        for li in range(len(w.lm_pos)):
            if li in inv and shared_joint[inv[li]] >= 0:
                m[li] = shared_joint[inv[li]]
                continue
            m[li] = len(lm_pos)
            lm_pos.append(w.lm_pos[li]); lm_slot.append(1)
            if li in inv:
                shared_joint[inv[li]] = m[li]
        lm_map.append(m)
    lm_slot = np.arange(len(lm_pos), dtype=np.int32)
    obs_lm = np.concatenate([m[w.obs_lm] for w, m in zip(wins, lm_map)]).astype(np.int32)
    obs_px = np.concatenate([w.obs_px for w in wins])
    off, pe = [0], []
    for w, eo in zip(wins, e_off):
        for o in range(len(w.obs_lm)):
            pe.extend(int(c) + 2 * eo for c in w.path_edge[w.obs_path_off[o]:w.obs_path_off[o + 1]])
            off.append(len(pe))
    return Window(np.concatenate(slots), np.concatenate(poses), lm_slot, np.array(lm_pos), obs_lm, obs_px, off, pe), e_off, lm_map, shared_joint
