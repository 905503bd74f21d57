"""GPU parity: persistent-workgroup LM kernel vs the CPU oracle, per LM iteration, within the tolerance
BASELINE.json's north_star states: pose / landmark updates within 1e-6 relative per LM iteration."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import synth
import synth_lm as sl

pytestmark = pytest.mark.gpu

REL = 1e-6     # north_star tolerance, per iteration


def to_capi(prm):
    from ssk_amd import capi
    out = capi.LmParams()
    C.memmove(C.byref(out), C.byref(prm), C.sizeof(out))
    return out


def rel_err(a, b, floor=0.0):
    """|a - b| / max(|b|, floor).  `floor`: the round-off level of the solve.  An LM run ends with attempts whose update is 1e-9 of
    the first one or less (the stop rule is a 1e-9 relative error decrease): such an update is -H^-1 g with a gradient that has
    cancelled to round-off, so it differs from kernel to kernel by the summation order of g (1e-5 of a 1e-10 m step was seen between
    two accumulation orders on the SAME device code).  The 1e-6 bar therefore applies to max(|update|, 1e-6 x the largest update of the
    run): an absolute 1e-12 of the run's largest step."""
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), floor, 1e-300))


def check_against_oracle(ctx, oracle, win, prm, min_accept=1):
    from ssk_amd import capi
    exp = oracle.lm_solve(win, prm)
    b = capi.LmBatch(ctx, [win], trace=True)
    b.run(to_capi(prm))
    res = b.results()[0]
    ent, dp, df = b.trace(0)
    poses, lms = b.state(0)
    er = exp["res"]
    assert res.iters == er.iters and res.stop_reason == er.stop_reason, (res.iters, er.iters, res.stop_reason, er.stop_reason)
    # the attempt that declares convergence changes the error by <= 1e-9 relative: whether that last step
    # is kept (err_new < err) is a coin toss at rounding level, every earlier decision must agree
    last_free = res.stop_reason == 4
    assert abs(res.accepted - er.accepted) <= (1 if last_free else 0)
    assert res.accepted >= min_accept and res.n_obs == er.n_obs
    assert abs(res.err_init - er.err_init) <= 1e-10 * er.err_init
    ok_it = [i for i, e in enumerate(exp["trace"]) if e.cholesky_ok]
    fl_p = 1e-6 * max([np.linalg.norm(exp["dp"][i]) for i in ok_it] + [0.0]) if win.E else 0.0
    fl_f = 1e-6 * max([np.linalg.norm(exp["df"][i]) for i in ok_it] + [0.0]) if win.L else 0.0
    for i, (g, e) in enumerate(zip(ent, exp["trace"])):
        assert g.cholesky_ok == e.cholesky_ok, i
        if not (last_free and i == res.iters - 1):
            assert g.accepted == e.accepted, i
        assert abs(g.lambda_ - e.lambda_) <= 1e-9 * abs(e.lambda_), (i, g.lambda_, e.lambda_)
        assert abs(g.err_after - e.err_after) <= REL * abs(e.err_after), (i, g.err_after, e.err_after)
        if e.cholesky_ok:
            if win.E:
                assert rel_err(dp[i], exp["dp"][i], fl_p) <= REL, (i, "edge update", rel_err(dp[i], exp["dp"][i]))
            if win.L:
                assert rel_err(df[i], exp["df"][i], fl_f) <= REL, (i, "landmark update", rel_err(df[i], exp["df"][i]))
    tol = 1e-5 if last_free else 1e-8
    # (absolute, except for a landmark the solve sends away -- depth unobservable, |f| ~ 1e7 --: relative to its size beyond 1 km)
    assert np.abs(poses - exp["poses"]).max() <= tol and (np.abs(lms - exp["lms"]) / np.where(np.abs(exp["lms"]) > 1e3, np.abs(exp["lms"]), 1.0)).max() <= 10 * tol
    assert abs(res.err_final - er.err_final) <= REL * er.err_final
    b.close()
    return res


@pytest.mark.parametrize("seed,robust", [(1, True), (2, False), (3, True)])
def test_lm_small_windows_per_iteration(ctx, oracle, seed, robust):
    win, gt = sl.make_submap_problem(seed, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    check_against_oracle(ctx, oracle, win, sl.make_params(robust=robust, sensor_t=(0.2, -0.05, 0.1)), min_accept=2)


def test_lm_config3_shape(ctx, oracle):
    """BASELINE.json configs[2] shape: submap_size 10, max_tree_depth 4, max_optimize_depth 5 (KITTI.ini:104-106),
    ~250 observations per keyframe, sigma 0.5 px, pseudo-Huber 1.5."""
    win, gt = sl.make_submap_problem(3, n_kf=40, n_lm=1400, submap_size=10, obs_per_kf=250)
    assert win.E >= 10 and len(win.obs_lm) >= 2000
    res = check_against_oracle(ctx, oracle, win, sl.make_params(), min_accept=2)
    assert 0.5 < res.err_final / res.n_obs < 1.5


def test_lm_noise_free_converges_to_ground_truth(ctx):
    from ssk_amd import capi
    win, gt = sl.make_submap_problem(4, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60, px_noise=0.0,
                                     edge_noise=(0.08, 0.015), lm_noise=0.08)
    b = capi.LmBatch(ctx, [win])
    b.run(to_capi(sl.make_params(max_iters=40, robust=False)))
    r = b.results()[0]
    poses, lms = b.state(0)
    unk = win.edge_unknown_slot >= 0
    assert r.err_final < 1e-6 * r.err_init
    assert np.abs(poses[unk] - gt["gt_pose"][unk]).max() < 1e-4 and np.abs(lms - gt["gt_lm"]).max() < 2e-3
    assert np.array_equal(poses[~unk], win.edge_inv_pose[~unk])
    b.close()


def test_lm_stage1_edges_only_and_empty(ctx, oracle):
    win, gt = sl.make_submap_problem(6, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    w1 = sl.Window(np.where(win.edge_unknown_slot == win.E - 1, 0, -1), win.edge_inv_pose,
                   -np.ones(len(win.lm_pos), np.int32), gt["gt_lm"], win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    check_against_oracle(ctx, oracle, w1, sl.make_params())
    from ssk_amd import capi
    w0 = sl.Window(-np.ones(len(win.edge_unknown_slot), np.int32), win.edge_inv_pose, -np.ones(len(win.lm_pos), np.int32),
                   win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    b = capi.LmBatch(ctx, [w0])
    b.run(to_capi(sl.make_params()))
    r = b.results()[0]
    assert r.stop_reason == 3 and r.iters == 0 and r.err_final == r.err_init
    assert abs(r.err_init - oracle.lm_total_error(w0, sl.make_params())) <= 1e-10 * r.err_init
    b.close()


@pytest.mark.parametrize("n_free,robust", [(1, True), (1, False), (2, True), (2, False)])
def test_lm_pose_only_windows_per_iteration(ctx, oracle, n_free, robust):
    """windows that optimise one or two edges and no landmark (stage 1 of define_new_keyframe: the new edge(s), every landmark held) run on
    lm_pose_only_kernel (no node tables): per iteration against the oracle, with and without the robust kernel, paths of several edges through the
    optimised ones in both directions; in a batch beside a window of another family; reset + rerun gives the same bits"""
    from ssk_amd import capi
    win, gt = sl.make_submap_problem(31 + n_free, n_kf=16, n_lm=260, submap_size=5, obs_per_kf=70)
    slots = np.full(len(win.edge_unknown_slot), -1, np.int32)
    order = np.argsort(-win.edge_unknown_slot)[:n_free]            # the newest edges
    for k, e in enumerate(sorted(order)):
        slots[e] = k
    w1 = sl.Window(slots, win.edge_inv_pose, -np.ones(len(win.lm_pos), np.int32), gt["gt_lm"], win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    assert w1.E == n_free and w1.L == 0
    b = capi.LmBatch(ctx, [w1])
    assert b.kernel_name() == "lm_pose_only_kernel"
    b.close()
    prm = sl.make_params(robust=robust)
    res = check_against_oracle(ctx, oracle, w1, prm)
    assert res.err_final <= res.err_init
    other = sl.make_submap_problem(5, n_kf=12, n_lm=200, submap_size=5, obs_per_kf=50)[0]
    b = capi.LmBatch(ctx, [other, w1, w1])
    b.run(to_capi(prm))
    r1 = b.results(); s1 = b.state(1)
    b.reset(); b.run(to_capi(prm))
    r2 = b.results(); s2 = b.state(2)
    assert (r1[1].iters, r1[1].err_final) == (res.iters, res.err_final) == (r2[2].iters, r2[2].err_final)
    assert np.array_equal(s1[0], s2[0])
    exp = oracle.lm_solve(other, prm, trace=False)
    assert r1[0].iters == exp["res"].iters
    b.close()


def test_lm_batch_of_ragged_windows_and_reset(ctx, oracle):
    """several windows of different sizes in one launch; reset + rerun reproduces the result bit for bit
    (deterministic summation order, no float atomics)."""
    from ssk_amd import capi
    wins = [sl.make_submap_problem(10 + i, n_kf=8 + 3 * i, n_lm=150 + 60 * i, submap_size=4 + i, obs_per_kf=40 + 10 * i)[0] for i in range(5)]
    prm = sl.make_params()
    b = capi.LmBatch(ctx, wins)
    b.run(to_capi(prm))
    res1 = b.results()
    st1 = [b.state(i) for i in range(len(wins))]
    for i, w in enumerate(wins):
        exp = oracle.lm_solve(w, prm, trace=False)
        assert res1[i].iters == exp["res"].iters and abs(res1[i].accepted - exp["res"].accepted) <= 1
        assert abs(res1[i].err_final - exp["res"].err_final) <= REL * exp["res"].err_final
        assert np.abs(st1[i][0] - exp["poses"]).max() <= 1e-5 and np.abs(st1[i][1] - exp["lms"]).max() <= 1e-4
    b.reset()
    b.run(to_capi(prm))
    res2 = b.results()
    for i in range(len(wins)):
        p2, l2 = b.state(i)
        assert np.array_equal(p2, st1[i][0]) and np.array_equal(l2, st1[i][1]) and res2[i].err_final == res1[i].err_final
    b.close()


@pytest.mark.parametrize("seed,robust", [(1, True), (2, False)])
def test_lm_general_kernel_same_contract(ctx, oracle, seed, robust, monkeypatch):
    """the scratch-in-L2 kernel (windows of any size; SSK_LM_KERNEL=l2 forces it) honours the same per-iteration bar
    as the LDS-tiled kernel the other tests run, on a small and on a multi-tile window."""
    monkeypatch.setenv("SSK_LM_KERNEL", "l2")
    win, gt = sl.make_submap_problem(seed, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    check_against_oracle(ctx, oracle, win, sl.make_params(robust=robust, sensor_t=(0.2, -0.05, 0.1)), min_accept=2)
    win, gt = sl.make_submap_problem(7, n_kf=60, n_lm=1500, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250,
                                     lm_fwd=(3, 20), z_vis=(2.5, 22))
    check_against_oracle(ctx, oracle, win, sl.make_params(robust=robust), min_accept=2)


def test_lm_tiled_kernel_multi_tile_window(ctx, oracle):
    """a bench-shaped window (~1.4k observations): several LDS tiles, segmented partial sums across tiles."""
    win, gt = sl.make_submap_problem(7, n_kf=60, n_lm=1500, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250,
                                     lm_fwd=(3, 20), z_vis=(2.5, 22))
    assert len(win.obs_lm) > 800
    check_against_oracle(ctx, oracle, win, sl.make_params(), min_accept=2)


@pytest.mark.parametrize("kernel", ["tiled", "l2", "node"])
def test_lm_large_window_global_cholesky(ctx, oracle, kernel, monkeypatch):
    """a window whose reduced system (6E > 132) does not fit LDS: the tiled kernel keeps it in L2 scratch (workgroup-wide
    Cholesky), as the general kernel and the BIG form of the node kernel do."""
    if kernel != "tiled":
        monkeypatch.setenv("SSK_LM_KERNEL", kernel)
    win, gt = sl.make_submap_problem(21, n_kf=45, n_lm=900, submap_size=3, max_tree_depth=6, opt_depth=40, obs_per_kf=60)
    assert 6 * win.E > 140
    check_against_oracle(ctx, oracle, win, sl.make_params(max_iters=8))


@pytest.mark.parametrize("n_kf", [24, 27, 41, 59])
@pytest.mark.parametrize("kernel", ["tiled", "coop", "node"])
def test_lm_dense_solver_beyond_lds_sizes(ctx, oracle, n_kf, kernel, monkeypatch):
    """ldl_solve_hybrid at the sizes where its paths change (6E = 138, 156: the whole packed triangle in LDS, no leading columns; 240: eleven leading
    six-column steps in L2 scratch beside the LDS block; 348: more leading columns than LDS rows, the back-substitution stages the rows in several
    chunks): every attempt against the oracle on one workgroup and on the cooperative kernel."""
    monkeypatch.setenv("SSK_LM_KERNEL", kernel)
    win, gt = sl.make_submap_problem(21, n_kf=n_kf, n_lm=20 * n_kf, submap_size=3, max_tree_depth=6, opt_depth=40, obs_per_kf=60)
    assert win.E == n_kf - 1 and 6 * win.E > 132
    res = check_against_oracle(ctx, oracle, win, sl.make_params(max_iters=8))
    assert res.stop_reason != 5


@pytest.mark.parametrize("kernel", ["tiled", "coop", "node"])
def test_lm_window_beyond_the_matrix_core_mask(ctx, oracle, kernel, monkeypatch):
    """E = 90 optimised edges: 6E / 16 = 34 row tiles of the reduced system, more than the 32-bit tile mask of the matrix-core Schur form holds --
    the library must fall back to the sparse form there (and say which kernel it runs), with the usual per-iteration bar."""
    from ssk_amd import capi
    monkeypatch.setenv("SSK_LM_KERNEL", kernel)
    win, gt = sl.make_submap_problem(23, n_kf=91, n_lm=1500, submap_size=3, max_tree_depth=6, opt_depth=90, obs_per_kf=40)
    assert win.E == 90
    b = capi.LmBatch(ctx, [win])
    assert "mfma" not in b.kernel_name() and b.kernel_name().startswith("lm_solve_" + kernel)
    b.close()
    res = check_against_oracle(ctx, oracle, win, sl.make_params(max_iters=5))
    assert res.stop_reason != 5


def test_lm_kernel_name_follows_the_batch(ctx, monkeypatch):
    """ssk_lm_batch_kernel_name: the node form for windows whose reduced system fits LDS, the tiled / cooperative kernels beyond, the forced ones."""
    from ssk_amd import capi
    small = sl.make_submap_problem(1, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)[0]
    large = sl.make_submap_problem(21, n_kf=45, n_lm=900, submap_size=3, max_tree_depth=6, opt_depth=40, obs_per_kf=60)[0]
    small10 = sl.make_submap_problem(6, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60, opt_depth=2)[0]
    assert small.E > 10 and 0 < small10.E <= 10
    for wins, env, expect in (([small], None, "lm_solve_node_kernel"), ([small10], None, "lm_solve_node_e10_kernel"), ([small, small10], None, "lm_solve_node_kernel"),
                              ([small], "tiled", "lm_solve_tiled"), ([large], None, "lm_split_linearise_kernel"),
                              ([large] * 128, None, "lm_split_linearise_kernel"), ([large], "coop", "lm_solve_coop"), ([large] * 128, "tiled", "lm_solve_tiled_kernel"),
                              ([large], "node", "lm_solve_node_big_kernel"), ([small], "split", "lm_split_linearise_kernel"), ([small, large], None, "lm_split_linearise_kernel"),
                              ([small], "l2", "lm_solve_kernel")):
        if env:
            monkeypatch.setenv("SSK_LM_KERNEL", env)
        else:
            monkeypatch.delenv("SSK_LM_KERNEL", raising=False)
        b = capi.LmBatch(ctx, wins)
        assert b.kernel_name().startswith(expect), (env, b.kernel_name())
        b.close()


def test_lm_set_state_equals_a_batch_built_on_that_state(ctx):
    """ssk_lm_batch_set_state: a batch whose structures were built on other initial values (the facade prepares the local-area window
    beside the solve of the new edges) and then given the final ones solves bit for bit like a batch built on them -- also after a reset."""
    import copy
    from ssk_amd import capi
    win, gt = sl.make_submap_problem(4, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    prm = to_capi(sl.make_params())
    ref = capi.LmBatch(ctx, [win]); ref.run(prm); r0 = ref.results()[0]; p0, l0 = ref.state(0); ref.close()
    stale = copy.copy(win)
    poses = np.array(win.edge_inv_pose, dtype=np.float64).reshape(-1, 12)
    lms = np.array(win.lm_pos, dtype=np.float64).reshape(-1, 3)
    stale.edge_inv_pose = (poses + 0.05).reshape(np.shape(win.edge_inv_pose))
    stale.lm_pos = (lms * 1.1).reshape(np.shape(win.lm_pos))
    b = capi.LmBatch(ctx, [stale])
    b.set_state(0, poses, lms)
    for _ in range(2):
        b.run(prm)
        r = b.results()[0]
        p, l = b.state(0)
        assert (r.iters, r.accepted, r.err_init, r.err_final) == (r0.iters, r0.accepted, r0.err_init, r0.err_final)
        assert np.array_equal(p, p0) and np.array_equal(l, l0)
        b.reset()
    b.close()


def run_plain(ctx, wins, prm):
    """No trace: (results, [(poses, lms)]) of a batch."""
    from ssk_amd import capi
    b = capi.LmBatch(ctx, wins)
    b.run(to_capi(prm))
    res = b.results()
    st = [b.state(i) for i in range(len(wins))]
    b.close()
    return res, st


def coop_cases(shape):
    if shape == "bench":
        wins = [sl.make_submap_problem(7, n_kf=60, n_lm=1500, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250,
                                       lm_fwd=(3, 20), z_vis=(2.5, 22))[0]]
        prm = sl.make_params()
    elif shape == "large":
        wins = [sl.make_submap_problem(21, n_kf=45, n_lm=900, submap_size=3, max_tree_depth=6, opt_depth=40, obs_per_kf=60)[0]]
        assert 6 * wins[0].E > 140                       # reduced system in L2 scratch: shared assembly, global-memory LDLT
        prm = sl.make_params(max_iters=8)
    else:
        wins = [sl.make_submap_problem(30 + k, n_kf=40 + 10 * k, n_lm=1000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=200,
                                       lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for k in range(3)]
        prm = sl.make_params()
    return wins, prm


@pytest.mark.parametrize("shape", ["bench", "large", "batch3"])
def test_lm_cooperative_kernel(ctx, oracle, shape, monkeypatch):
    """lm_solve_coop_kernel (several workgroups per window, barriers in global memory; SSK_LM_KERNEL=coop forces it, the
    default picks it for few large windows) -- the kernel the un-batched drop-in runs: the same per-iteration bar against the
    ORACLE as every other kernel (lambda, error, edge and landmark updates of every attempt within 1e-6, every decision but
    the documented last-step coin toss identical), then the final state against the one-workgroup kernel."""
    wins, prm = coop_cases(shape)
    monkeypatch.setenv("SSK_LM_KERNEL", "coop")
    for w in wins:
        res = check_against_oracle(ctx, oracle, w, prm)
        assert res.stop_reason != 5
    resg, stg = run_plain(ctx, wins, prm)
    monkeypatch.setenv("SSK_LM_KERNEL", "tiled")
    res1, st1 = run_plain(ctx, wins, prm)
    for w, (r1, rg) in enumerate(zip(res1, resg)):
        assert rg.stop_reason == r1.stop_reason != 5 and rg.iters == r1.iters and rg.n_obs == r1.n_obs
        assert abs(rg.err_final - r1.err_final) <= 1e-9 * r1.err_final
        tol = 1e-5 if rg.stop_reason == 4 else 1e-8
        assert np.abs(stg[w][0] - st1[w][0]).max() <= tol and np.abs(stg[w][1] - st1[w][1]).max() <= 10 * tol


def test_lm_cooperative_barrier_timeout_and_fallback(ctx, oracle, monkeypatch):
    """The cooperative kernel's barriers are bounded waits: with SSK_LM_COOP_TEST_TIMEOUT every barrier waits for an arrival
    that never comes (what a workgroup sees when a co-tenant of the GPU keeps a partner from being scheduled).  Without the
    fallback the windows end in stop_reason 5 and ssk_lm_batch_results reports it; with it (the default) the library goes
    back to the state the launch started from and repeats the solve with one workgroup per window."""
    from ssk_amd import capi
    wins, prm = coop_cases("bench")
    monkeypatch.setenv("SSK_LM_KERNEL", "tiled")
    res1, st1 = run_plain(ctx, wins, prm)
    monkeypatch.setenv("SSK_LM_KERNEL", "coop")
    monkeypatch.setenv("SSK_LM_COOP_TEST_TIMEOUT", "1")
    monkeypatch.setenv("SSK_LM_COOP_NO_FALLBACK", "1")
    b = capi.LmBatch(ctx, wins)
    b.run(to_capi(prm))
    with pytest.raises(capi.SskError) as ei:
        b.results()
    assert "did not all arrive" in str(ei.value)
    raw = b.results(check=False)
    assert raw[0].stop_reason == 5
    b.close()
    monkeypatch.delenv("SSK_LM_COOP_NO_FALLBACK")
    b = capi.LmBatch(ctx, wins)
    b.run(to_capi(prm))
    res = b.results()                                   # timed out, restored, repeated by lm_solve_tiled_kernel
    assert res[0].stop_reason == res1[0].stop_reason != 5 and res[0].iters == res1[0].iters and res[0].err_final == res1[0].err_final
    p, l = b.state(0)
    assert np.array_equal(p, st1[0][0]) and np.array_equal(l, st1[0][1])
    b.close()


def test_lm_golden_case_on_gpu(ctx):
    """tests/golden/lm_case0.npz (window + the oracle's per-attempt errors and final state, committed): the GPU against the
    fixture itself, no oracle library in the loop."""
    import os
    from ssk_amd import capi
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lm_case0.npz"))
    win = sl.Window(g["edge_unknown_slot"], g["edge_inv_pose"], g["lm_unknown_slot"], g["lm_pos"], g["obs_lm"], g["obs_px"],
                    g["obs_path_off"], g["path_edge"])
    b = capi.LmBatch(ctx, [win], trace=True)
    b.run(to_capi(sl.make_params()))
    res = b.results()[0]
    ent, dp, df = b.trace(0)
    poses, lms = b.state(0)
    assert res.iters == int(g["iters"]) and abs(res.accepted - int(g["accepted"])) <= (1 if res.stop_reason == 4 else 0)
    assert np.allclose([t.err_after for t in ent], g["err_after"], rtol=REL, atol=0)
    tol = 1e-5 if res.stop_reason == 4 else 1e-8
    assert np.abs(poses - g["final_poses"]).max() <= tol and np.abs(lms - g["final_lms"]).max() <= 10 * tol
    b.close()


@pytest.mark.parametrize("seed", range(8))
def test_lm_bench_windows_against_oracle(ctx, oracle, seed, monkeypatch):
    """The exact windows bench.py times (BASELINE configs[2]: 200 keyframes / 5000 landmarks, submap 10, tree depth 4, window =
    the 10 most recent edges; seeds 0-7 on rank 0), per iteration against the oracle, on the kernel the bench's batch runs them on (the node
    kernel; a batch of a few such windows would go to the phase-split kernels: SSK_LM_FEW_SPLIT=0 keeps the batch rule) -- and, for two seeds,
    on the phase-split kernels a single window of this size takes by default."""
    from ssk_amd import capi
    win = sl.make_submap_problem(seed, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2,
                                 obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
    assert win.E == 10 and len(win.obs_lm) > 1000
    if seed < 2:
        b = capi.LmBatch(ctx, [win])
        assert b.kernel_name() == "lm_split_linearise_kernel"
        b.close()
        res = check_against_oracle(ctx, oracle, win, sl.make_params(), min_accept=2)
        assert res.stop_reason in (1, 4) and res.err_final < res.err_init
    monkeypatch.setenv("SSK_LM_FEW_SPLIT", "0")
    b = capi.LmBatch(ctx, [win])
    assert b.kernel_name() == "lm_solve_node_e10_kernel"
    b.close()
    res = check_against_oracle(ctx, oracle, win, sl.make_params(), min_accept=2)
    assert res.stop_reason in (1, 4) and res.err_final < res.err_init


def test_lm_bench_windows_seeds_8_to_63(ctx, oracle):
    """bench.py's default line solves 64 distinct windows (seeds 0-63 on rank 0); seeds 0-7 are checked per iteration above, the other 56 here as
    ONE batch (the way the bench runs them): iteration count, stop reason, final error and final state of every window against the oracle."""
    from ssk_amd import capi
    wins = [sl.make_submap_problem(seed, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
            for seed in range(8, 64)]
    prm = sl.make_params()
    b = capi.LmBatch(ctx, wins)
    assert b.kernel_name() == "lm_solve_node_e10_kernel"       # (every window E = 10: the variant whose dense solver is the one-wavefront matrix-core ldl_solve_mfma_wave)
    b.run(to_capi(prm))
    res = b.results()
    for i, w in enumerate(wins):
        exp = oracle.lm_solve(w, prm, trace=False)
        er = exp["res"]
        assert res[i].iters == er.iters and res[i].stop_reason == er.stop_reason, (i + 8, res[i].iters, er.iters, res[i].stop_reason, er.stop_reason)
        assert abs(res[i].accepted - er.accepted) <= (1 if er.stop_reason == 4 else 0)
        assert abs(res[i].err_init - er.err_init) <= 1e-10 * er.err_init and abs(res[i].err_final - er.err_final) <= REL * er.err_final
        p, l = b.state(i)
        tol = 1e-5 if er.stop_reason == 4 else 1e-8
        assert np.abs(p - exp["poses"]).max() <= tol and np.abs(l - exp["lms"]).max() <= 10 * tol, i + 8
    b.close()


@pytest.mark.parametrize("kernel", ["tiled", "coop"])
def test_lm_sparse_schur_form(ctx, oracle, kernel, monkeypatch):
    """SSK_LM_SCHUR=sparse: the Schur complement as sparse 6x6 items Y_i Y_j^T with segmented lane sums (the form every other test
    runs with the matrix cores: v_mfma_f64_16x16x4_f64 on Y panels) -- same per-iteration bar, one workgroup per window and the
    cooperative kernel, on a bench-shaped window and on one whose reduced system lives in L2 scratch."""
    monkeypatch.setenv("SSK_LM_SCHUR", "sparse")
    monkeypatch.setenv("SSK_LM_KERNEL", kernel)
    for shape in ("bench", "large"):
        wins, prm = coop_cases(shape)
        res = check_against_oracle(ctx, oracle, wins[0], prm)
        assert res.stop_reason != 5


@pytest.mark.parametrize("schur", ["mfma", "sparse"])
def test_lm_on_chip_accumulation_kernel(ctx, oracle, schur, monkeypatch):
    """lm_solve_acc_kernel / lm_solve_acc_mfma_kernel (SSK_LM_KERNEL=acc forces them; the default for batches of windows with
    E <= 12 optimised edges): H_pp blocks, reduced gradient and the Schur complement accumulate in LDS -- the Schur complement
    either as sparse 6x6 items (SSK_LM_SCHUR=sparse) or as v_mfma_f64_16x16x4_f64 products of Y panels (=mfma, default).  Same
    per-iteration bar against the oracle on a one-tile window, a stage-1 window (edges only), bench-shaped multi-tile windows
    and a window at the E = 12 limit; deterministic run to run."""
    from ssk_amd import capi
    monkeypatch.setenv("SSK_LM_KERNEL", "acc")
    monkeypatch.setenv("SSK_LM_SCHUR", schur)
    win, gt = sl.make_submap_problem(1, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    check_against_oracle(ctx, oracle, win, sl.make_params(robust=True, sensor_t=(0.2, -0.05, 0.1)), min_accept=2)
    w1 = sl.Window(np.where(win.edge_unknown_slot == win.E - 1, 0, -1), win.edge_inv_pose,
                   -np.ones(len(win.lm_pos), np.int32), gt["gt_lm"], win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    check_against_oracle(ctx, oracle, w1, sl.make_params())
    for seed in (7, 8):
        win = sl.make_submap_problem(seed, n_kf=60, n_lm=1500, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250,
                                     lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
        assert win.E == 10 and len(win.obs_lm) > 800
        check_against_oracle(ctx, oracle, win, sl.make_params(robust=seed == 7), min_accept=2)
    win = sl.make_submap_problem(5, n_kf=60, n_lm=1500, submap_size=12, max_tree_depth=4, opt_depth=2, obs_per_kf=200, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
    assert win.E == 12, win.E
    check_against_oracle(ctx, oracle, win, sl.make_params(), min_accept=2)
    # a batch of ragged windows, twice: bit-identical (fixed order of every accumulator's adds)
    wins = [sl.make_submap_problem(10 + i, n_kf=8 + 3 * i, n_lm=150 + 60 * i, submap_size=4 + i, obs_per_kf=40 + 10 * i, opt_depth=2)[0] for i in range(5)]
    assert all(w.E <= 12 for w in wins)
    b = capi.LmBatch(ctx, wins)
    b.run(to_capi(sl.make_params()))
    r1 = b.results(); s1 = [b.state(i) for i in range(len(wins))]
    b.reset(); b.run(to_capi(sl.make_params()))
    r2 = b.results()
    for i, w in enumerate(wins):
        exp = oracle.lm_solve(w, sl.make_params(), trace=False)
        assert r1[i].iters == exp["res"].iters and abs(r1[i].err_final - exp["res"].err_final) <= REL * exp["res"].err_final
        p2, l2 = b.state(i)
        assert np.array_equal(p2, s1[i][0]) and np.array_equal(l2, s1[i][1]) and r2[i].err_final == r1[i].err_final
    b.close()


DEPTH5 = dict(n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))


def stage1_window(w):
    """stage 1 of define_new_keyframe (EST.cpp:782-784): the newest edge alone, landmarks held -- bench.py's lm_depth5 stage-1 window"""
    newest = int(np.flatnonzero(w.edge_unknown_slot == w.E - 1)[0])
    keep = [o for o in range(len(w.obs_lm)) if w.obs_path_off[o + 1] > w.obs_path_off[o] and (w.path_edge[w.obs_path_off[o]] >> 1) == newest]
    off, pe = [0], []
    for o in keep:
        pe.extend(w.path_edge[w.obs_path_off[o]:w.obs_path_off[o + 1]]); off.append(len(pe))
    return sl.Window(np.where(w.edge_unknown_slot == w.E - 1, 0, -1), w.edge_inv_pose, -np.ones(len(w.lm_pos), np.int32), w.lm_pos,
                     w.obs_lm[keep], w.obs_px[keep], off, pe)


@pytest.mark.parametrize("seed", range(4))
@pytest.mark.parametrize("kernel", [None, "tiled", "coop"])
def test_lm_depth5_bench_windows_per_iteration(ctx, oracle, seed, kernel, monkeypatch):
    """The four windows bench.py's `lm_depth5` object times (the reference's real per-keyframe LM load, KITTI.ini:104-106: optimise depth 5 ->
    40 edges, 8.3 k observations), at their size, per iteration against the oracle, on every kernel that can serve them: the default (the
    phase-split kernels, ssk_lm_split.h), one workgroup per window (SSK_LM_KERNEL=tiled) and the cooperative kernel."""
    from ssk_amd import capi
    if kernel:
        monkeypatch.setenv("SSK_LM_KERNEL", kernel)
    win = sl.make_submap_problem(seed, **DEPTH5)[0]
    assert win.E == 40 and len(win.obs_lm) > 8000
    b = capi.LmBatch(ctx, [win])
    assert b.kernel_name().startswith({None: "lm_split_", "tiled": "lm_solve_tiled", "coop": "lm_solve_coop"}[kernel]), b.kernel_name()
    b.close()
    res = check_against_oracle(ctx, oracle, win, sl.make_params(), min_accept=2)
    assert res.stop_reason in (0, 1, 4) and res.err_final < res.err_init
    if kernel is None:
        check_against_oracle(ctx, oracle, stage1_window(win), sl.make_params())


@pytest.mark.parametrize("seed,robust", [(1, True), (2, False), (3, True)])
def test_lm_split_small_windows_per_iteration(ctx, oracle, seed, robust, monkeypatch):
    """the phase-split kernels forced (SSK_LM_KERNEL=split) on windows the node kernel would take: one contract, whatever the size"""
    monkeypatch.setenv("SSK_LM_KERNEL", "split")
    win, gt = sl.make_submap_problem(seed, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    check_against_oracle(ctx, oracle, win, sl.make_params(robust=robust, sensor_t=(0.2, -0.05, 0.1)), min_accept=2)
    # stage 1 (one edge, landmarks held), landmarks only, a third of the landmarks held
    w1 = sl.Window(np.where(win.edge_unknown_slot == win.E - 1, 0, -1), win.edge_inv_pose,
                   -np.ones(len(win.lm_pos), np.int32), gt["gt_lm"], win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    check_against_oracle(ctx, oracle, w1, sl.make_params(robust=robust))
    w0 = sl.Window(-np.ones(len(win.edge_unknown_slot), np.int32), win.edge_inv_pose, win.lm_unknown_slot, win.lm_pos,
                   win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    check_against_oracle(ctx, oracle, w0, sl.make_params(robust=robust))
    s2 = win.lm_unknown_slot.copy(); s2[::3] = -1; s2[s2 >= 0] = np.arange((s2 >= 0).sum())
    w2 = sl.Window(win.edge_unknown_slot, win.edge_inv_pose, s2, win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    check_against_oracle(ctx, oracle, w2, sl.make_params(robust=robust))


def test_lm_split_bench_window_and_empty(ctx, oracle, monkeypatch):
    from ssk_amd import capi
    monkeypatch.setenv("SSK_LM_KERNEL", "split")
    win = sl.make_submap_problem(0, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
    check_against_oracle(ctx, oracle, win, sl.make_params(), min_accept=2)
    w0 = sl.Window(-np.ones(len(win.edge_unknown_slot), np.int32), win.edge_inv_pose, -np.ones(len(win.lm_pos), np.int32),
                   win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    b = capi.LmBatch(ctx, [w0])
    b.run(to_capi(sl.make_params()))
    r = b.results()[0]
    assert r.stop_reason == 3 and r.iters == 0 and r.err_final == r.err_init
    assert abs(r.err_init - oracle.lm_total_error(w0, sl.make_params())) <= 1e-10 * r.err_init
    b.close()
    # max_iters = 0: nothing but the initial error
    b = capi.LmBatch(ctx, [win])
    b.run(to_capi(sl.make_params(max_iters=0)))
    r = b.results()[0]
    assert r.iters == 0 and r.stop_reason == 0 and r.err_final == r.err_init
    b.close()


@pytest.mark.parametrize("n_kf", [24, 38, 41, 43, 45, 56, 59, 63, 67, 91])
def test_lm_split_dense_solver_sizes(ctx, oracle, n_kf):
    """the default path of windows beyond LDS at the sizes where the dense solvers' paths change: 6E = 138 (nine tile rows of the MFMA-tile solver,
    the border row inside the last), 222 (border row 14 of the last tile row), 240 (border row = a tile row of its own), 252 (the largest system
    held WHOLE in accumulator tiles); round 5: 264, 330, 348, 372, 396 ON THE MATRIX-CORE PATH too (ldl_solve_mfma_lead: 16 .. 144 leading columns by
    16-column panels of accumulator tiles -- ldl_leading_mfma --, their updates applied to the trailing tiles as MFMAs) -- the local areas the
    facade's own stream grows reach 61 edges (94 at KITTI-00 length); 540: 288 leading columns, 32 staged columns to a chunk step"""
    win, gt = sl.make_submap_problem(21 if n_kf < 91 else 23, n_kf=n_kf, n_lm=(20 * n_kf if n_kf < 91 else 1500), submap_size=3, max_tree_depth=6,
                                     opt_depth=n_kf, obs_per_kf=60 if n_kf < 91 else 40)
    assert win.E == n_kf - 1
    from ssk_amd import capi
    b = capi.LmBatch(ctx, [win])
    assert b.kernel_name() == "lm_split_linearise_kernel"
    # (every window the node builder takes is solved on the matrix cores: within 255 unknowns the tile solver alone -- its own instance of K2 --,
    # beyond them the leading columns first)
    assert b.solver_name() == ("ldl_solve_mfma_tiles" if 6 * win.E <= 255 else "ldl_solve_mfma_lead")
    b.close()
    check_against_oracle(ctx, oracle, win, sl.make_params(max_iters=8 if n_kf < 91 else 5))


def test_lm_split_solver_and_table_memory_switches(ctx, oracle, monkeypatch):
    """the fallbacks of the default path give the default path's results: SSK_K2_SOLVER=regtile (the register-tile solver for a depth-5 local area
    instead of the MFMA-tile solver: the same update to rounding level, the same verdicts) and SSK_LM_NO_PINNED=1 (the batch's tables in pageable
    memory instead of the context's pinned arena: the same bits)"""
    from ssk_amd import capi
    prm = sl.make_params()
    wins = [sl.make_submap_problem(1, **DEPTH5)[0]] + [sl.make_submap_problem(20 + i, n_kf=10 + 2 * i, n_lm=200 + 50 * i, submap_size=5, obs_per_kf=50)[0] for i in range(3)]
    def solve():
        b = capi.LmBatch(ctx, wins)
        b.run(to_capi(prm))
        r = b.results()
        st = [b.state(i) for i in range(len(wins))]
        b.close()
        return r, st
    r0, s0 = solve()
    monkeypatch.setenv("SSK_LM_NO_PINNED", "1")
    r1, s1 = solve()
    monkeypatch.delenv("SSK_LM_NO_PINNED")
    for i in range(len(wins)):
        assert (r0[i].iters, r0[i].accepted, r0[i].err_final) == (r1[i].iters, r1[i].accepted, r1[i].err_final)
        assert np.array_equal(s0[i][0], s1[i][0]) and np.array_equal(s0[i][1], s1[i][1])
    monkeypatch.setenv("SSK_K2_SOLVER", "regtile")
    r2, s2 = solve()
    for i in range(len(wins)):
        assert r0[i].iters == r2[i].iters and abs(r0[i].accepted - r2[i].accepted) <= 1
        assert abs(r0[i].err_final - r2[i].err_final) <= REL * r0[i].err_final
        assert np.abs(s0[i][0] - s2[i][0]).max() <= 1e-6


def test_lm_split_mixed_batch_is_reproducible(ctx, oracle):
    """windows of very different sizes in ONE batch (a depth-5 local area among bench windows and small ones: no all-or-nothing kernel choice):
    every window against the oracle; reset + rerun and the same windows in another batch order give the same bits (one owner and a fixed order
    per sum: a window's result does not depend on the batch around it)."""
    from ssk_amd import capi
    wins = [sl.make_submap_problem(10 + i, n_kf=8 + 3 * i, n_lm=150 + 60 * i, submap_size=4 + i, obs_per_kf=40 + 10 * i)[0] for i in range(4)]
    wins.append(sl.make_submap_problem(1, **DEPTH5)[0])
    wins.append(sl.make_submap_problem(2, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0])
    prm = sl.make_params()
    b = capi.LmBatch(ctx, wins)
    assert b.kernel_name() == "lm_split_linearise_kernel"
    b.run(to_capi(prm))
    res1 = b.results()
    st1 = [b.state(i) for i in range(len(wins))]
    for i, w in enumerate(wins):
        exp = oracle.lm_solve(w, prm, trace=False)
        assert res1[i].iters == exp["res"].iters and abs(res1[i].accepted - exp["res"].accepted) <= 1
        assert abs(res1[i].err_final - exp["res"].err_final) <= REL * exp["res"].err_final
        assert np.abs(st1[i][0] - exp["poses"]).max() <= 1e-5 and np.abs(st1[i][1] - exp["lms"]).max() <= 1e-4
    b.reset()
    b.run(to_capi(prm))
    res2 = b.results()
    for i in range(len(wins)):
        p2, l2 = b.state(i)
        assert np.array_equal(p2, st1[i][0]) and np.array_equal(l2, st1[i][1]) and res2[i].err_final == res1[i].err_final
    b.close()
    b = capi.LmBatch(ctx, wins[::-1])
    b.run(to_capi(prm))
    res3 = b.results()
    for i in range(len(wins)):
        p3, l3 = b.state(len(wins) - 1 - i)
        assert np.array_equal(p3, st1[i][0]) and np.array_equal(l3, st1[i][1]) and res3[len(wins) - 1 - i].err_final == res1[i].err_final
    b.close()


def test_lm_k2_instance_does_not_change_a_windows_bits(ctx, oracle):
    """K2 has an instance per batch shape (round 5): the tile solver alone when every window of the batch has at most 255 unknowns, leading columns +
    tiles as soon as one window is larger.  A depth-5 local area (240 unknowns) and a 13-edge window give the same bits next to a 70-edge window
    (420 unknowns: the second instance) as in a batch of their own (the first), and the large window agrees with the oracle."""
    from ssk_amd import capi
    prm = sl.make_params(max_iters=8)
    small = [sl.make_submap_problem(1, **DEPTH5)[0], sl.make_submap_problem(13, n_kf=14, n_lm=400, submap_size=5, max_tree_depth=6, opt_depth=14, obs_per_kf=60)[0]]
    big = sl.make_submap_problem(0, **dict(DEPTH5, opt_depth=8))[0]
    assert 6 * big.E > 255 and all(6 * w.E <= 255 for w in small)
    b = capi.LmBatch(ctx, small)
    assert b.kernel_name() == "lm_split_linearise_kernel" and b.solver_name() == "ldl_solve_mfma_tiles"
    b.run(to_capi(prm))
    r0 = b.results(); st0 = [b.state(i) for i in range(2)]
    b.close()
    b = capi.LmBatch(ctx, [big] + small)
    assert b.solver_name() == "ldl_solve_mfma_lead"
    b.run(to_capi(prm))
    r1 = b.results()
    for i in range(2):
        p, l = b.state(1 + i)
        assert r1[1 + i].iters == r0[i].iters and r1[1 + i].err_final == r0[i].err_final
        assert np.array_equal(p, st0[i][0]) and np.array_equal(l, st0[i][1])
    exp = oracle.lm_solve(big, prm, trace=False)
    pb, lb = b.state(0)
    assert r1[0].iters == exp["res"].iters and abs(r1[0].err_final - exp["res"].err_final) <= REL * exp["res"].err_final
    assert np.abs(pb - exp["poses"]).max() <= 1e-5 and np.abs(lb - exp["lms"]).max() <= 1e-4
    b.close()


@pytest.mark.parametrize("E", [12, 15, 18])
def test_lm_node_kernel_middle_branch(ctx, oracle, E, monkeypatch):
    """lm_solve_node_kernel beyond the one-wavefront block solver of the E <= 10 variant: the 4x4 register tiles (E = 12: 10 < E <= 14) and, between
    them and the BIG form, the dense copy out of the block accumulators + ldl_solve_lds at 256 threads (E = 15, 18) -- the size of the facade's
    local areas at depth 3-5 early in a sequence.  With trace, robust on / off and a window whose landmarks are partly held."""
    from ssk_amd import capi
    monkeypatch.setenv("SSK_LM_KERNEL", "node")
    win, gt = sl.make_submap_problem(21, n_kf=E + 1, n_lm=20 * (E + 1), submap_size=3, max_tree_depth=6, opt_depth=40, obs_per_kf=60)
    assert win.E == E
    b = capi.LmBatch(ctx, [win])
    assert b.kernel_name() == "lm_solve_node_kernel"
    b.close()
    for robust in (True, False):
        check_against_oracle(ctx, oracle, win, sl.make_params(max_iters=10, robust=robust))
    s2 = win.lm_unknown_slot.copy(); s2[::4] = -1; s2[s2 >= 0] = np.arange((s2 >= 0).sum())
    w2 = sl.Window(win.edge_unknown_slot, win.edge_inv_pose, s2, win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
    check_against_oracle(ctx, oracle, w2, sl.make_params(max_iters=10))


def test_build_observations_bit_exact(ctx, oracle):
    """a10/a11: projectMatchTo3D + camera pose composition (UTIL.h:558-574, EST.cpp:158,698), FP64 bit-exact."""
    import ctypes
    rng = np.random.default_rng(5)
    kf = synth.synth_kf(rng, 400, 400, 300)
    kf["kp_right"]["x"] = np.floor(rng.uniform(31, 600, 400))
    kf["kp_left"]["x"] = np.floor(rng.uniform(640, 1200, 400))       # positive disparities
    prm = sl.make_params(sensor_t=(0.27, 0.0, 1.65))
    px, pos = ctx.build_observations(kf["kp_left"], kf["kp_right"], kf["matches"], to_capi(prm))
    epx = np.zeros_like(px)
    epos = np.zeros_like(pos)
    kl = np.ascontiguousarray(kf["kp_left"]); kr = np.ascontiguousarray(kf["kp_right"]); m = np.ascontiguousarray(kf["matches"])
    oracle.lib.ssko_build_observations(ctypes.c_void_p(kl.ctypes.data), ctypes.c_void_p(kr.ctypes.data), ctypes.c_void_p(m.ctypes.data),
                                       len(m), ctypes.byref(prm), ctypes.c_void_p(epx.ctypes.data), ctypes.c_void_p(epos.ctypes.data))
    assert np.array_equal(px, epx) and np.array_equal(pos, epos)


def test_build_observations_batch_equals_the_host_entry(ctx, oracle):
    """ssk_build_observations_batch (a10 / a11 for a resident keyframe batch, device to device) against the per-keyframe host entry on the same
    keyframes: pixels and initial landmark positions bit for bit."""
    import torch
    from ssk_amd import capi
    B = 3
    kb = capi.KfBatch(ctx, B, synth.W, synth.H, 512)
    for k in range(B):
        l, r = synth.synth_stereo_pair(40 + k)
        kb.upload(k, l, r)
    kb.run_frontend(capi.FrontendParams(**synth.KITTI_FRONTEND))
    prm = to_capi(sl.make_params(sensor_t=(0.27, 0.0, 1.65)))
    dev = torch.device("cuda", 0)
    d_px = torch.zeros(B, 512, 4, dtype=torch.float32, device=dev)
    d_pos = torch.zeros(B, 512, 3, dtype=torch.float64, device=dev)
    kb.build_observations(prm, d_px, d_pos)
    ctx.synchronize()
    px, pos = d_px.cpu().numpy(), d_pos.cpu().numpy()
    for k in range(B):
        kf = kb.download(k)
        m = kf["matches"]
        assert len(m) > 100
        epx, epos = ctx.build_observations(kf["kp_left"], kf["kp_right"], m, prm)
        assert np.array_equal(px[k, :len(m)], epx) and np.array_equal(pos[k, :len(m)], epos)
        assert not pos[k, len(m):].any()
    kb.close()


def _window_with_oversize_landmark(win, min_obs=70):
    """the window with the observations of its most-observed landmark repeated (pixels moved by a fraction of a pixel): a landmark of more than 64
    observations -- beyond a wavefront's lanes, the node builder's limit (ssk_lm_node_build.h)"""
    cnt = np.bincount(win.obs_lm, minlength=len(win.lm_pos))
    l = int(np.argmax(cnt))
    times = -(-min_obs // int(cnt[l]))
    sel = np.flatnonzero(win.obs_lm == l)
    obs_lm, obs_px = [win.obs_lm], [win.obs_px]
    off, pe = list(win.obs_path_off), list(win.path_edge)
    rng = np.random.default_rng(5)
    for _ in range(times - 1):
        obs_lm.append(win.obs_lm[sel]); obs_px.append((win.obs_px[sel] + rng.normal(0, 0.2, (len(sel), 4))).astype(np.float32))
        for o in sel:
            pe.extend(win.path_edge[win.obs_path_off[o]:win.obs_path_off[o + 1]]); off.append(len(pe))
    w2 = sl.Window(win.edge_unknown_slot, win.edge_inv_pose, win.lm_unknown_slot, win.lm_pos, np.concatenate(obs_lm), np.concatenate(obs_px), off, pe)
    return w2, int(cnt[l]) * times


def test_lm_one_oversize_window_in_a_batch_keeps_the_others_on_their_kernels(ctx, oracle):
    """Round 5 (VERDICT r4 'weak' 1): the lean builder was all-or-nothing -- ONE window with a landmark of more than 64 observations sent the WHOLE batch
    to the rounds-1-2 kernels.  Now that window alone goes through the full builder (a sub-batch), the 63 others keep lm_solve_node_e10_kernel:
    kernel name, every window against the oracle, the others' bits unchanged (round 6: the fitting windows' tables are built ONCE and the small-batch
    decisions are taken on the whole batch's size, so they cannot change kernels), and a loose bound on the batch's time."""
    import time
    from ssk_amd import capi
    wins = [sl.make_submap_problem(seed, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
            for seed in range(64)]
    odd, n_obs_l = _window_with_oversize_landmark(wins[17], min_obs=70)
    assert n_obs_l > 64
    mixed = list(wins); mixed[17] = odd
    prm = sl.make_params()
    def run(batch_wins):
        b = capi.LmBatch(ctx, batch_wins)
        name = b.kernel_name()
        ts = []
        for _ in range(4):
            b.reset(); ctx.synchronize()
            t0 = time.perf_counter(); b.run(to_capi(prm)); res = b.results(); ts.append(time.perf_counter() - t0)
        states = [b.state(i) for i in (0, 17, 63)]
        b.close()
        return name, res, min(ts[1:]), states
    name0, res0, t_plain, _ = run(wins)
    name1, res1, t_mixed, st1 = run(mixed)
    assert name0 == "lm_solve_node_e10_kernel"
    assert name1.startswith("lm_solve_node_e10_kernel+lm_solve"), name1          # the 63 on the node kernel + the odd one on a kernel of the full builder
    for i, w in enumerate(mixed):
        er = oracle.lm_solve(w, prm, trace=False)["res"]
        r = res1[i]
        assert r.iters == er.iters and r.stop_reason == er.stop_reason and r.n_obs == er.n_obs, (i, r.iters, er.iters)
        assert abs(r.err_init - er.err_init) <= 1e-10 * er.err_init and abs(r.err_final - er.err_final) <= REL * er.err_final, i
        if i != 17:                                                                # the windows beside it: the SAME bits as in the plain batch
            assert r.err_final == res0[i].err_final and r.iters == res0[i].iters, i
    exp = oracle.lm_solve(odd, prm, trace=False)
    assert np.abs(st1[1][0] - exp["poses"]).max() <= 1e-5 and np.abs(st1[1][1] - exp["lms"]).max() <= 1e-4
    # (a wall-clock ratio on a shared box: the bar is loose -- the kernel names and the bit equality above are the test; measured 1.2-1.3 x)
    assert t_mixed <= 3.0 * t_plain, (t_mixed, t_plain)
    print("64 bench windows: %.3f ms; with one window of a %d-observation landmark (%s): %.3f ms" % (1e3 * t_plain, n_obs_l, name1, 1e3 * t_mixed))


def test_lm_large_batches_created_back_to_back_and_run_through_another_context(ctx, pkg):
    """Round 6: a large lean batch's ssk_lm_batch_create returns with its uploads (from the context's pinned arena) still queued; the context's NEXT
    create waits for them before it writes the arena, and a batch that only ever ran through ANOTHER context is destroyed without draining the creating
    context's stream.  Three 128-window batches (~15 MB of tables each: the table-by-table upload path) created back to back by one context, run through
    another, must give the results of a batch created and run the ordinary way -- bit for bit -- and so must a batch created after one was destroyed."""
    from ssk_amd import capi
    wins = [sl.make_submap_problem(seed, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]
            for seed in range(8)]
    sets = [[wins[(k + s) % 8] for k in range(128)] for s in range(3)]
    prm = to_capi(sl.make_params())

    def outcome(b, run_ctx=None):
        b.run(prm, ctx=run_ctx)
        if run_ctx is not None:
            run_ctx.synchronize()
        return [(r.iters, r.stop_reason, r.accepted, r.err_init, r.err_final) for r in b.results()]

    ref = []
    for s in range(3):
        b = capi.LmBatch(ctx, sets[s])
        ref.append(outcome(b))
        b.close()
    ctx_c = pkg.Context(0)
    made = [capi.LmBatch(ctx_c, sets[s]) for s in range(3)]          # back to back: each create finds the previous one's uploads pending
    for s in (2, 0, 1):
        assert outcome(made[s], run_ctx=ctx) == ref[s], s
    made[1].close()
    again = capi.LmBatch(ctx_c, sets[1])                              # (takes the freed slab while the other two batches are alive)
    assert outcome(again, run_ctx=ctx) == ref[1]
    for b in (made[0], made[2], again):
        b.close()
    ctx_c.close()
