"""GPU parity: fundamental-matrix RANSAC filter (fm_kernel) vs the CPU oracle through the C-ABI.  Both sides run the same
IEEE operation sequence (no contraction, no libm in the loop) with the same counter-based generator, so the model, the
inlier set, the verdicts and the hypothesis counts are compared bit for bit."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
import synth
import synth_cip as sc

pytestmark = pytest.mark.gpu


def fm_params(**over):
    from ssk_amd import capi
    o = ol.kitti_fm_params(**over)
    c = capi.FmParams()
    C.memmove(C.byref(c), C.byref(o), C.sizeof(c))
    return c, o


def check(ctx, oracle, this_xy, other_xy, train_idx, status, **over):
    cp, op = fm_params(**over)
    g_st, g = ctx.fm_filter(this_xy, other_xy, train_idx, status, cp)
    e_st, e = oracle.fundamental_filter(this_xy, other_xy, train_idx, status, op)
    assert (g.valid, g.n_pairs, g.n_inliers, g.n_rejected, g.hypotheses, g.best_hypothesis) == \
           (e.valid, e.n_pairs, e.n_inliers, e.n_rejected, e.hypotheses, e.best_hypothesis)
    assert np.array_equal(np.array(g.F[:]), np.array(e.F[:])) and np.array_equal(g_st, e_st)
    return g, g_st


@pytest.mark.parametrize("seed,over", [(1, {}), (2, dict(seed=12345)), (5, dict(threshold=1.0, confidence=0.99)), (6, dict(max_iters=5)),
                                       (7, dict(threshold=0.5))])
def test_fm_matches_oracle_bit_for_bit(ctx, oracle, seed, over):
    c = sc.make_case(seed)
    check(ctx, oracle, c["this_obs"][:, :2], c["other_obs"][:, :2], c["train_idx"], c["status"], **over)


def test_fm_many_outliers_runs_several_rounds(ctx, oracle):
    """40 % gross outliers: the adaptive stop needs more than one round of 64 hypotheses."""
    c = sc.make_case(8, n_this=400, n_other=420, n_outliers=160, frac_pre_rejected=0.0)
    g, st = check(ctx, oracle, c["this_obs"][:, :2], c["other_obs"][:, :2], c["train_idx"], c["status"])
    assert g.valid == 1 and g.hypotheses > 64
    assert (st[c["outlier"]] == 4).mean() > 0.9


def test_fm_edge_cases(ctx, oracle):
    c = sc.make_case(3, n_this=30, n_other=40, n_outliers=0)
    st0 = np.ones(30, np.int32); st0[:14] = 0
    g, st = check(ctx, oracle, c["this_obs"][:, :2], c["other_obs"][:, :2], c["train_idx"], st0)      # 14 pairs < min_points
    assert g.valid == 0 and np.array_equal(st, st0)
    check(ctx, oracle, c["this_obs"][:, :2], c["other_obs"][:, :2], c["train_idx"], np.ones(30, np.int32))   # nothing tracked
    same = np.tile(np.array([[300.0, 200.0]], np.float32), (20, 1))
    check(ctx, oracle, same, same, np.arange(20, dtype=np.int32), np.zeros(20, np.int32))             # degenerate: every sample is singular
    big = sc.make_case(9, n_this=2000, n_other=2048, n_outliers=200)
    check(ctx, oracle, big["this_obs"][:, :2], big["other_obs"][:, :2], big["train_idx"], big["status"])


def test_da_fundamental_filter_drop_in(ctx, oracle):
    from ssk_amd import capi
    c = sc.make_case(10, n_this=300, n_other=320)

    def kf(obs):
        n = len(obs)
        kl = np.zeros(n, capi.KEYPOINT_DTYPE); kr = np.zeros(n, capi.KEYPOINT_DTYPE)
        kl["x"], kl["y"], kr["x"], kr["y"] = obs[:, 0], obs[:, 1], obs[:, 2], obs[:, 3]
        m = np.zeros(n, capi.DMATCH_DTYPE); m["queryIdx"] = np.arange(n); m["trainIdx"] = np.arange(n)
        return capi.make_kf_view(kl, kr, np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8), m)
    (va, _ka), (vb, _kb) = kf(c["this_obs"]), kf(c["other_obs"])
    da = np.zeros(len(c["train_idx"]), capi.DMATCH_DTYPE)
    da["queryIdx"] = np.arange(len(da)); da["trainIdx"] = c["train_idx"]
    cp, op = fm_params()
    st1, o1 = ctx.da_fundamental_filter(va, vb, da, c["status"], cp)
    st2, o2 = ctx.fm_filter(c["this_obs"][:, :2], c["other_obs"][:, :2], c["train_idx"], c["status"], cp)
    assert np.array_equal(st1, st2) and np.array_equal(np.array(o1.F[:]), np.array(o2.F[:]))


def test_fm_batched_on_frontend_outputs(ctx, oracle, pkg):
    """ssk_fm_run_batch on device-resident keyframes: pair p behaves as the host entry with seed + p."""
    import torch
    from ssk_amd import capi
    B, N = 3, 512
    cur, prev = capi.KfBatch(ctx, B, synth.W, synth.H, N), capi.KfBatch(ctx, B, synth.W, synth.H, N)
    frames, _ = synth.synth_translation_sequence(3, B + 1)          # a camera moving sideways: real parallax between keyframes
    for i in range(B):                                              # (a pure image shift would make every 7-point sample singular)
        cur.upload(i, *frames[i + 1])
        prev.upload(i, *frames[i])
    fp = capi.FrontendParams(**synth.KITTI_FRONTEND)
    cur.run_frontend(fp); prev.run_frontend(fp)
    pairs = np.stack([np.arange(B), np.arange(B)], 1).astype(np.int32)
    dev = torch.device("cuda", 0)
    d_idx = torch.empty(B * N, dtype=torch.int32, device=dev); d_dist = torch.empty_like(d_idx); d_st = torch.empty_like(d_idx)
    d_tr = torch.empty(B, dtype=torch.int32, device=dev)
    ctx.da_run_batch(cur, prev, pairs, capi.DaParams(1, 1, synth.H, 60.0), d_idx, d_dist, d_st, d_tr)
    ctx.synchronize(); torch.cuda.synchronize()
    st0, idx0 = d_st.cpu().numpy().reshape(B, N).copy(), d_idx.cpu().numpy().reshape(B, N).copy()
    cp, op = fm_params()
    ctx.fm_run_batch(cur, prev, pairs, cp, d_idx, d_st, d_tr)
    ctx.synchronize(); torch.cuda.synchronize()
    st1, tr1 = d_st.cpu().numpy().reshape(B, N), d_tr.cpu().numpy()
    valid = 0
    for i in range(B):
        a, b = cur.download(i), prev.download(i)
        xy = lambda k: np.stack([k["kp_left"]["x"][k["matches"]["queryIdx"]], k["kp_left"]["y"][k["matches"]["queryIdx"]]], 1).astype(np.float32)
        n = len(a["matches"])
        _cp, op_i = fm_params(seed=synth.KITTI_FM["seed"] + i)
        e_st, e = oracle.fundamental_filter(xy(a), xy(b), idx0[i, :n], st0[i, :n], op_i)
        assert e.n_pairs >= 100
        assert np.array_equal(st1[i, :n], e_st) and int(tr1[i]) == int((e_st == 0).sum())
        valid += e.valid
    assert valid >= 1
    cur.close(); prev.close()
