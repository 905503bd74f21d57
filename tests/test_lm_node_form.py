"""The NODE form of the reduced camera system (srba-stereo-slam_amd/csrc/ssk_lm_node.h) on the CPU: the symbolic structure the library's
host builder produces (ssk_lm_node_build.h) is driven through a CPU emulation of the kernel's data flow (tools/lm_node_emu.cpp, compiled
here with g++) and compared with the dense textbook form (J^T J, Schur complement, written out in numpy below: contract L1-L5 of
oracle/oracle_lm.c) -- reduced matrix, reduced gradient, landmark step, the lambda_0 pass.  No GPU: guards the tiler and the bit fields."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import synth_lm as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    out = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, "liblm_node_emu.so")
    src = os.path.join(ROOT, "tools", "lm_node_emu.cpp")
    deps = [src] + [os.path.join(ROOT, "srba-stereo-slam_amd", "csrc", f) for f in ("ssk_lm_node.h", "ssk_lm_node_build.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", src, "-o", lib], check=True)
    return C.CDLL(lib)


def skew(q):
    return np.array([[0, -q[2], q[1]], [q[2], 0, -q[0]], [-q[1], q[0], 0.0]])


def dense(win, prm, lam):
    """Textbook form: per observation the chain, the 4x6 / 4x3 Jacobian blocks, H_pp, H_ff, W; Schur complement; steps."""
    K = sl.KITTI_CAM
    S = np.array(prm.sensor_pose).reshape(3, 4)
    E, L = win.E, win.L
    n = 6 * E
    Hpp = np.zeros((n, n)); gp = np.zeros(n); Hff = np.zeros((L, 3, 3)); gf = np.zeros((L, 3)); Wpf = np.zeros((n, 3 * L))
    kp = prm.kernel_param
    for o in range(len(win.obs_lm)):
        p0, p1 = win.obs_path_off[o], win.obs_path_off[o + 1]
        npth = p1 - p0
        q = [None] * (npth + 1)
        q[npth] = win.lm_pos[win.obs_lm[o]].copy()
        for i in range(npth - 1, -1, -1):
            c = win.path_edge[p0 + i]; T = win.edge_inv_pose[c >> 1].reshape(3, 4)
            q[i] = T[:, :3] @ q[i + 1] + T[:, 3] if (c & 1) == 0 else T[:, :3].T @ (q[i + 1] - T[:, 3])
        RA = [np.eye(3)]
        for i in range(npth):
            c = win.path_edge[p0 + i]; T = win.edge_inv_pose[c >> 1].reshape(3, 4)
            RA.append(RA[i] @ (T[:, :3] if (c & 1) == 0 else T[:, :3].T))
        pc = S[:, :3].T @ (q[0] - S[:, 3])
        iz = 1 / pc[2]
        h = np.array([K['cx'] + K['fx'] * pc[0] * iz, K['cy'] + K['fy'] * pc[1] * iz, K['cx'] + K['fx'] * (pc[0] - K['baseline']) * iz, K['cy'] + K['fy'] * pc[1] * iz])
        r = win.obs_px[o].astype(float) - h
        s2 = r @ r
        w = np.sqrt(1 / np.sqrt(1 + s2 / kp ** 2)) if prm.use_robust_kernel else 1.0
        D = np.array([[K['fx'] * iz, 0, -K['fx'] * pc[0] * iz * iz], [0, K['fy'] * iz, -K['fy'] * pc[1] * iz * iz],
                      [K['fx'] * iz, 0, -K['fx'] * (pc[0] - K['baseline']) * iz * iz], [0, K['fy'] * iz, -K['fy'] * pc[1] * iz * iz]])
        M = D @ S[:, :3].T
        Jf = w * M @ RA[npth]; wr = w * r
        ls = win.lm_unknown_slot[win.obs_lm[o]]
        Je = []
        for i in range(npth):
            c = win.path_edge[p0 + i]; es = win.edge_unknown_slot[c >> 1]
            if es < 0:
                continue
            d = c & 1
            RR = RA[i] if d == 0 else RA[i + 1]; qq = q[i] if d == 0 else q[i + 1]; sg = w if d == 0 else -w
            A = sg * M @ RR
            Je.append((es, np.hstack([A, -A @ skew(qq)])))
        if ls >= 0:
            Hff[ls] += Jf.T @ Jf; gf[ls] += Jf.T @ wr
        for ei, Ji in Je:
            gp[6 * ei:6 * ei + 6] += Ji.T @ wr
            if ls >= 0:
                Wpf[6 * ei:6 * ei + 6, 3 * ls:3 * ls + 3] += Ji.T @ Jf
            for ej, Jj in Je:
                Hpp[6 * ei:6 * ei + 6, 6 * ej:6 * ej + 6] += Ji.T @ Jj
    Hred = Hpp + lam * np.eye(n); gred = gp.copy()
    Vinv = np.zeros((L, 3, 3))
    for l in range(L):
        Vinv[l] = np.linalg.inv(Hff[l] + lam * np.eye(3))
        Wl = Wpf[:, 3 * l:3 * l + 3]
        Hred -= Wl @ Vinv[l] @ Wl.T
        gred -= Wl @ Vinv[l] @ gf[l]
    dp = np.linalg.solve(Hred, gred) if n else np.zeros(0)
    df = np.array([Vinv[l] @ (gf[l] - Wpf[:, 3 * l:3 * l + 3].T @ dp) for l in range(L)]).reshape(L, 3)
    return Hred, gred, dp, df, Hpp, Hff


def run_emu(emu, win, prm, lam, pre=False, dp=None, waves=4, form=0):
    n = 6 * win.E
    H = np.zeros((n, n)); g = np.zeros(n + 1); df = np.zeros((max(win.L, 1), 3)); vmax = C.c_double(0); stats = (C.c_int * 5)()
    wc = win.c()
    rc = emu.nd_emu_linearise_form(C.byref(wc), C.byref(prm), C.c_double(lam), int(pre), waves, H.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p),
                                   dp.ctypes.data_as(C.c_void_p) if dp is not None else None, df.ctypes.data_as(C.c_void_p) if dp is not None else None,
                                   C.byref(vmax), stats, form)
    assert rc == 0
    return H, g[:n], df[:win.L], vmax.value, list(stats)


def check(emu, win, prm, lam=0.37, waves=4, forms=(0, 1)):
    """form 0: the tables of the persistent node kernel; form 1: those of the phase-split kernels (landmarks clustered by edge set, one
    partial-sum slot per (tile, block), block-major reduction)"""
    ref = dense(win, prm, lam)
    for form in forms:
        stats = check_form(emu, win, prm, lam, waves, form, ref)
    return stats


def check_form(emu, win, prm, lam, waves, form, ref):
    H1, g1, dp1, df1, Hpp, Hff = ref
    H2, g2, df2, _, stats = run_emu(emu, win, prm, lam, dp=np.ascontiguousarray(dp1), waves=waves, form=form)
    if win.E:
        sc = np.abs(H1).max()
        assert np.abs(H1 - H2).max() <= 1e-12 * sc, np.abs(H1 - H2).max() / sc
        assert np.abs(g1 - g2).max() <= 1e-12 * np.abs(g1).max()
    if win.L:
        assert np.abs(df1 - df2).max() <= 1e-9 * max(np.abs(df1).max(), 1e-300)
    # the lambda_0 pass: diagonal blocks of H_pp alone, largest diagonal entry of an optimised landmark's block
    H3, _, _, vmax, _ = run_emu(emu, win, prm, 0.0, pre=True, waves=waves, form=form)
    for e in range(win.E):
        blk = slice(6 * e, 6 * e + 6)
        assert np.abs(np.triu(H3[blk, blk]) - np.triu(Hpp[blk, blk])).max() <= 1e-12 * np.abs(Hpp).max()
    if win.L:
        assert abs(vmax - max(Hff[l][a, a] for l in range(win.L) for a in range(3))) <= 1e-12 * vmax
    return stats


def test_node_form_small_windows(emu):
    for seed, kw in [(1, dict(n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)), (6, dict(n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60, opt_depth=2))]:
        win, gt = sl.make_submap_problem(seed, **kw)
        for robust in (True, False):
            check(emu, win, sl.make_params(robust=robust, sensor_t=(0.2, -0.05, 0.1)))
        # a third of the landmarks held (shared landmarks of the multi-GPU exchange; stage 1 of define_new_keyframe holds all of them)
        s2 = win.lm_unknown_slot.copy(); s2[::3] = -1; s2[s2 >= 0] = np.arange((s2 >= 0).sum())
        w2 = sl.Window(win.edge_unknown_slot, win.edge_inv_pose, s2, win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
        check(emu, w2, sl.make_params())
        w1 = sl.Window(np.where(win.edge_unknown_slot == win.E - 1, 0, -1), win.edge_inv_pose, -np.ones(len(win.lm_pos), np.int32), win.lm_pos,
                       win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
        check(emu, w1, sl.make_params())
        w0 = sl.Window(-np.ones(len(win.edge_unknown_slot), np.int32), win.edge_inv_pose, win.lm_unknown_slot, win.lm_pos,
                       win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)         # landmarks only
        check(emu, w0, sl.make_params())


def test_node_form_bench_window_and_wave_counts(emu):
    win = sl.make_submap_problem(0, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
    assert win.E == 10 and len(win.obs_lm) > 1000
    for waves in (4, 8):
        stats = check(emu, win, sl.make_params(), waves=waves)
        assert stats[0] == (stats[1] + waves - 1) // waves and stats[2] > 4000


def test_split_form_depth5_window(emu):
    """The reference's real per-keyframe load (KITTI.ini:104-106: optimise depth 5 -> 40 edges, 8.3 k observations): the tables of the phase-split
    form against the dense textbook system; the clustering by edge set must cut the (tile, block) pairs well below the slot order's ~16 k."""
    win = sl.make_submap_problem(0, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
    assert win.E == 40 and len(win.obs_lm) > 8000
    stats = check(emu, win, sl.make_params(), forms=(1,))
    assert stats[2] > 40000 and stats[4] < 7000, stats


def test_node_form_two_nodes_of_a_landmark_on_one_edge(emu):
    """Paths that are not those of one spanning tree: some observations go forth and back over an extra edge, so a landmark gets two
    different suffixes that start with the same optimised edge (two nodes on one edge, both orders of the pair in the diagonal block)
    and paths that cross an edge twice."""
    win, gt = sl.make_submap_problem(2, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    off, pe = [0], []
    rng = np.random.default_rng(5)
    n_mod = 0
    for o in range(len(win.obs_lm)):
        p = list(win.path_edge[win.obs_path_off[o]:win.obs_path_off[o + 1]])
        if len(p) in (1, 2) and rng.random() < 0.4:
            c = int(rng.integers(0, 2 * len(win.edge_unknown_slot)))
            p = p[:1] + [c, c ^ 1] + p[1:]          # detour after the first step: the same transform, another suffix
            n_mod += 1
        pe.extend(p); off.append(len(pe))
    assert n_mod > 50
    w2 = sl.Window(win.edge_unknown_slot, win.edge_inv_pose, win.lm_unknown_slot, win.lm_pos, win.obs_lm, win.obs_px, off, pe)
    check(emu, w2, sl.make_params())


def test_builder_tables_unchanged(emu):
    """The symbolic builder's tables (all 16, node and split form) against the golden hashes written with the round-5 builder
    (tests/golden/make_node_table_hashes.py): the round-6 rewrite of its loops -- per-thread workspace, rank sort, run-based nesting test,
    counting sorts, one sort key per landmark -- changes no byte of any table."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_node_table_hashes as g
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "node_table_hashes.json")))
    assert len(gold) == len(g.CASES)
    for case in g.CASES:
        got = g.hashes(emu, case)
        assert got == gold["%s/%d" % (case[0], case[1])], case[:2]
    # one window's landmark and tile ranges built by helper threads side by side (nd_team: the un-batched drop-in's single local area): the same bytes
    for case in g.CASES:
        if case[0] in ("d5", "est", "d2"):
            for threads in (2, 3, 8):
                assert g.hashes(emu, case, threads) == gold["%s/%d" % (case[0], case[1])], (case[:2], threads)
