"""GPU parity: change-in-pose outlier filter (cip_kernel) vs the CPU oracle through the C-ABI: identical verdicts
(integer work), pose / residuals within 1e-6 relative (the FP64 sums run in another order)."""
import numpy as np
import pytest

import oracle_lib as ol
import synth
import synth_cip as sc

pytestmark = pytest.mark.gpu
REL = 1e-6


def cip_params(**over):
    from ssk_amd import capi
    o = ol.kitti_cip_params(**over)
    import ctypes as C
    c = capi.CipParams()
    C.memmove(C.byref(c), C.byref(o), C.sizeof(c))
    return c, o


def check(ctx, oracle, case, init, **over):
    cp, op = cip_params(**over)
    g_st, g_res, g = ctx.cip_filter(case["this_obs"], case["other_obs"], case["train_idx"], case["status"], init, cp)
    e_st, e_res, e = oracle.change_in_pose(case["this_obs"], case["other_obs"], case["train_idx"], case["status"], init, op)
    assert g.valid == e.valid and g.n_pairs == e.n_pairs
    if e.valid:
        assert (g.iters_initial, g.iters_final, g.n_inliers) == (e.iters_initial, e.iters_final, e.n_inliers)
        assert np.abs(np.array(g.pose[:]) - np.array(e.pose[:])).max() <= REL
        assert np.allclose(g_res, e_res, rtol=REL, atol=1e-9)
        assert abs(g.cost - e.cost) <= REL * max(e.cost, 1.0)
        # a verdict may only differ where the residual sits on the gate to within the tolerance
        diff = g_st != e_st
        if diff.any():
            trk = np.flatnonzero(case["status"] == 0)
            r_of = dict(zip(trk, e_res))
            assert all(abs(r_of[k] - op.residual_th) <= REL * op.residual_th for k in np.flatnonzero(diff))
    else:
        assert np.array_equal(g_st, e_st)
    assert g.n_rejected == int(((g_st == 5) & (case["status"] == 0)).sum())
    return g, g_st


@pytest.mark.parametrize("seed,over", [(1, {}), (2, dict(use_robust_kernel=0)), (5, dict(kernel_param=1.0, residual_threshold=8.0)),
                                       (6, dict(initial_max_iters=3, max_iters=2))])
def test_cip_matches_oracle(ctx, oracle, seed, over):
    check(ctx, oracle, sc.make_case(seed), sc.identity_pose(), **over)


def test_cip_large_motion_and_warm_start(ctx, oracle):
    c = sc.make_case(7, w=(0.05, 0.12, -0.03), t=(0.4, -0.1, -2.5), n_outliers=60)
    g, st = check(ctx, oracle, c, sc.identity_pose())
    assert g.valid and np.all(st[(c["status"] == 0) & c["outlier"]] == 5)
    check(ctx, oracle, c, np.hstack([c["R"], c["t"][:, None]]).reshape(12))


def test_cip_edge_cases(ctx, oracle):
    c = sc.make_case(3, n_this=12, n_other=20, n_outliers=0)
    few = dict(c, status=np.r_[np.zeros(4, np.int32), np.ones(8, np.int32)])
    g, st = check(ctx, oracle, few, sc.identity_pose())                       # fewer than bad_tracking_th pairs: invalid, all rejected
    assert g.valid == 0 and np.all(st[:4] == 5)
    none = dict(c, status=np.ones(12, np.int32))
    g, st = check(ctx, oracle, none, sc.identity_pose())                      # nothing tracked
    assert g.valid == 0 and g.n_pairs == 0 and np.all(st == 1)
    big = sc.make_case(8, n_this=2000, n_other=2048, n_outliers=150)          # capacity: n_feats limit of the front-end
    check(ctx, oracle, big, sc.identity_pose())


def test_da_change_in_pose_drop_in(ctx, oracle):
    """the reference-shaped entry (keyframe views + this_matches) marshals exactly what ssk_cip_filter gets."""
    from ssk_amd import capi
    c = sc.make_case(9, n_this=300, n_other=320)

    def kf(obs):
        n = len(obs)
        kl = np.zeros(n, capi.KEYPOINT_DTYPE); kr = np.zeros(n, capi.KEYPOINT_DTYPE)
        kl["x"], kl["y"], kr["x"], kr["y"] = obs[:, 0], obs[:, 1], obs[:, 2], obs[:, 3]
        m = np.zeros(n, capi.DMATCH_DTYPE); m["queryIdx"] = np.arange(n); m["trainIdx"] = np.arange(n)
        return capi.make_kf_view(kl, kr, np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8), m)
    (va, _ka), (vb, _kb) = kf(c["this_obs"]), kf(c["other_obs"])
    da = np.zeros(len(c["train_idx"]), capi.DMATCH_DTYPE)
    da["queryIdx"] = np.arange(len(da)); da["trainIdx"] = c["train_idx"]
    cp, op = cip_params()
    st1, r1, o1 = ctx.da_change_in_pose(va, vb, da, c["status"], sc.identity_pose(), cp)
    st2, r2, o2 = ctx.cip_filter(c["this_obs"], c["other_obs"], c["train_idx"], c["status"], sc.identity_pose(), cp)
    assert np.array_equal(st1, st2) and np.array_equal(r1, r2) and np.array_equal(np.array(o1.pose[:]), np.array(o2.pose[:]))


def test_cip_batched_on_frontend_outputs(ctx, oracle, pkg):
    """ssk_cip_run_batch on device-resident keyframes: the same verdicts as the host entry fed with the downloaded data."""
    import torch
    from ssk_amd import capi
    B, N = 3, 512
    cur, prev = capi.KfBatch(ctx, B, synth.W, synth.H, N), capi.KfBatch(ctx, B, synth.W, synth.H, N)
    for i in range(B):
        l, r = synth.synth_stereo_pair(40 + i)
        cur.upload(i, l, r)
        prev.upload(i, np.roll(l, (1, -2), (0, 1)), np.roll(r, (1, -2), (0, 1)))
    fp = capi.FrontendParams(**synth.KITTI_FRONTEND)
    cur.run_frontend(fp); prev.run_frontend(fp)
    pairs = np.stack([np.arange(B), np.arange(B)], 1).astype(np.int32)
    dev = torch.device("cuda", 0)
    d_idx = torch.empty(B * N, dtype=torch.int32, device=dev); d_dist = torch.empty_like(d_idx); d_st = torch.empty_like(d_idx)
    d_tr = torch.empty(B, dtype=torch.int32, device=dev)
    ctx.da_run_batch(cur, prev, pairs, capi.DaParams(1, 1, synth.H, 60.0), d_idx, d_dist, d_st, d_tr)
    ctx.synchronize(); torch.cuda.synchronize()
    st0, idx0 = d_st.cpu().numpy().reshape(B, N).copy(), d_idx.cpu().numpy().reshape(B, N).copy()
    cp, op = cip_params()
    ctx.cip_run_batch(cur, prev, pairs, cp, d_idx, d_st, d_tr)
    ctx.synchronize(); torch.cuda.synchronize()
    st1, tr1 = d_st.cpu().numpy().reshape(B, N), d_tr.cpu().numpy()
    for i in range(B):
        a, b = cur.download(i), prev.download(i)

        def obs(k):
            m = k["matches"]
            return np.stack([k["kp_left"]["x"][m["queryIdx"]], k["kp_left"]["y"][m["queryIdx"]],
                             k["kp_right"]["x"][m["trainIdx"]], k["kp_right"]["y"][m["trainIdx"]]], 1).astype(np.float32)
        n = len(a["matches"])
        e_st, e_res, e = oracle.change_in_pose(obs(a), obs(b), idx0[i, :n], st0[i, :n], sc.identity_pose(), op)
        assert e.valid == 1 and e.n_pairs >= 100
        diff = np.flatnonzero(st1[i, :n] != e_st)
        assert len(diff) <= 1, (i, diff)                                       # a residual exactly on the gate is the only freedom
        assert abs(int(tr1[i]) - int((e_st == 0).sum())) <= len(diff)
    cur.close(); prev.close()
