"""GPU parity: HIP data-association path (through the C-ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import oracle_lib as ol
import synth

pytestmark = pytest.mark.gpu


def prm_pair(pkg, vals):
    from ssk_amd import capi
    return capi.DaParams(*vals), ol.DaParams(*vals)


@pytest.mark.parametrize("nq,nt", [(1, 1), (63, 64), (64, 65), (65, 511), (500, 500), (500, 513), (37, 1100), (1100, 3)])
def test_hamming_host_vs_oracle(ctx, oracle, nq, nt):
    rng = np.random.default_rng(nq * 10007 + nt)
    q = synth.random_descriptors(rng, nq)
    t = synth.random_descriptors(rng, nt)
    # plant exact duplicates and near ties so the first-minimum rule is exercised
    for i in range(0, min(nq, nt), 7):
        t[(i * 3) % nt] = q[i]
        t[(i * 5 + 1) % nt] = q[i]
    got = ctx.hamming_match(q, t)
    exp = oracle.hamming_match(q, t)
    assert np.array_equal(got, exp)


def test_hamming_empty(ctx):
    z = np.zeros((0, 32), np.uint8)
    q = np.zeros((5, 32), np.uint8)
    assert len(ctx.hamming_match(q, z)) == 0 and len(ctx.hamming_match(z, q)) == 0


def test_hamming_ties_low_entropy(ctx, oracle):
    """descriptors drawn from a tiny alphabet: most minima are ties -> lowest train index must win."""
    rng = np.random.default_rng(5)
    base = synth.random_descriptors(rng, 6)
    q = base[rng.integers(0, 6, 300)]
    t = base[rng.integers(0, 6, 700)]
    assert np.array_equal(ctx.hamming_match(q, t), oracle.hamming_match(q, t))


def test_hamming_batched_dev_gather(ctx, oracle):
    """device-resident batched form with ragged pairs and the fused match-index gather (a5)."""
    import torch
    rng = np.random.default_rng(11)
    sizes = [(300, 280), (1, 500), (500, 1), (64, 64), (129, 513), (0, 10), (10, 0), (77, 640)]
    pool_q = synth.random_descriptors(rng, 2000)
    pool_t = synth.random_descriptors(rng, 2500)
    q_off, t_off, qg, tg = [0], [0], [], []
    for nq, nt in sizes:
        qg += list(rng.integers(0, len(pool_q), nq))
        tg += list(rng.integers(0, len(pool_t), nt))
        q_off.append(len(qg))
        t_off.append(len(tg))
    dev = torch.device("cuda:0")
    d = lambda a, dt: torch.from_numpy(np.asarray(a, dt)).to(dev)
    d_q, d_t = d(pool_q, np.uint8), d(pool_t, np.uint8)
    d_qo, d_to = d(q_off, np.int32), d(t_off, np.int32)
    d_qg, d_tg = d(qg, np.int32), d(tg, np.int32)
    d_idx = torch.full((len(qg),), -7, dtype=torch.int32, device=dev)
    d_dist = torch.full((len(qg),), -7, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ctx.hamming_match_batched_dev(d_q, d_qo, d_qg, d_t, d_to, d_tg, len(sizes), max(s[0] for s in sizes), d_idx, d_dist)
    ctx.synchronize()
    idx, dist = d_idx.cpu().numpy(), d_dist.cpu().numpy()
    for p, (nq, nt) in enumerate(sizes):
        qs = pool_q[np.asarray(qg[q_off[p]:q_off[p + 1]], int)] if nq else np.zeros((0, 32), np.uint8)
        ts = pool_t[np.asarray(tg[t_off[p]:t_off[p + 1]], int)] if nt else np.zeros((0, 32), np.uint8)
        exp = oracle.hamming_match(qs, ts)
        sl = slice(q_off[p], q_off[p + 1])
        if nt == 0:
            assert (idx[sl] == -1).all() and (dist[sl] == 0x7FFFFFFF).all()
        else:
            assert np.array_equal(idx[sl], exp["trainIdx"]) and np.array_equal(dist[sl], exp["distance"].astype(np.int32))


def test_direction_thresholds_are_libm_preimages(ctx):
    """The 19 thresholds must be the exact preimages of the reference formula's bin edges under THIS
    host's atanf: bin(T) == b and bin(prev float) == b-1."""
    th = ctx.direction_thresholds()
    # bin 18 starts where ang rounds to 180.0 in double: a hair below quotient 0
    assert np.all(np.diff(th) > 0) and -1e-15 < th[9] <= 0.0

    def ref_bin(q):
        ang = 180 / np.pi * float(np.arctan(np.float32(q)).astype(np.float32)) + 180.0
        return int(np.floor(ang / 10))
    # numpy's float32 arctan may differ from libm in the last ulp; only check well inside the bins here,
    # the exact libm comparison happens through the oracle in test_da_filters_threshold_edges.
    for b, t in zip(range(9, 28), th):
        if abs(t) < 100 and abs(t) > 1e-3:
            assert ref_bin(t * (1 + 1e-5) if t > 0 else t * (1 - 1e-5)) == b


def test_da_filters_threshold_edges(ctx, oracle, pkg):
    """quotients exactly on, just below and just above every bin edge: kernel bins == libm bins."""
    th = ctx.direction_thresholds()
    qs = []
    for t in th:
        if np.isfinite(t):
            qs += [t, np.nextafter(t, np.float32(-np.inf), dtype=np.float32), np.nextafter(t, np.float32(np.inf), dtype=np.float32)]
    qs += [np.float32(np.inf), np.float32(-3.0e38), np.float32(3.0e38), np.float32(-0.0), np.float32(1e-45)]
    qs = np.array(qs, np.float32)
    n = len(qs)
    # quotient = (oy + 0 - 0) / (1 - 0) = oy exactly; +inf via dx == 0
    this_xy = np.zeros((n, 2), np.float32)
    other_xy = np.stack([np.ones(n, np.float32), qs], 1)
    inf = np.isinf(qs)
    other_xy[inf, 0] = 0.0
    other_xy[inf, 1] = 5.0
    m = np.zeros(n, ol.DMATCH_DTYPE)
    m["queryIdx"] = np.arange(n)
    m["trainIdx"] = np.arange(n)
    m["distance"] = 10
    # run one query at a time against a crowd in a fixed bin so each status reveals that query's bin
    for crowd_bin_q in [np.float32(1.0), np.float32(-1.0), np.float32(0.05), np.float32(-0.05), np.float32(30.0), np.float32(-30.0)]:
        crowd = 40
        txy = np.zeros((n + crowd, 2), np.float32)
        oxy = np.vstack([other_xy, np.tile(np.array([[1.0, crowd_bin_q]], np.float32), (crowd, 1))])
        mm = np.zeros(n + crowd, ol.DMATCH_DTYPE)
        mm["queryIdx"] = np.arange(n + crowd)
        mm["trainIdx"] = np.arange(n + crowd)
        mm["distance"] = 10
        cp, op = prm_pair(pkg, (1, 0, 0, 60.0))
        got, gt = ctx.da_filters(mm, n + crowd, txy, oxy, cp)
        exp, et = oracle.da_filters(mm, n + crowd, txy, oxy, op)
        assert np.array_equal(got, exp) and gt == et


@pytest.mark.parametrize("seed,nf,nm,vals", [
    (1, 500, 320, (1, 1, 370, 60.0)), (2, 500, 499, (1, 1, 376, 60.0)), (3, 200, 100, (0, 1, 370, 60.0)),
    (4, 300, 150, (1, 0, 370, 60.0)), (5, 64, 30, (1, 1, 370, 20.0)), (6, 500, 250, (0, 0, 370, 60.0)),
    (7, 40, 1, (1, 1, 370, 60.0)), (8, 500, 400, (1, 1, 0, 255.0)),
])
def test_data_association_vs_oracle(ctx, oracle, pkg, seed, nf, nm, vals):
    from ssk_amd import capi
    rng = np.random.default_rng(seed)
    a, b = synth.synth_kf_pair(rng, nf, nm, noise_bits=20, outlier_frac=0.2)
    cp, op = prm_pair(pkg, vals)
    va, ka = capi.make_kf_view(**a)
    vb, kb = capi.make_kf_view(**b)
    oa, oka = ol.make_kf_view(**a)
    ob, okb = ol.make_kf_view(**b)
    got_m, got_s, got_t = ctx.data_association(va, vb, cp)
    exp_m, exp_s, exp_t = oracle.data_association(oa, ob, op)
    assert np.array_equal(got_m, exp_m)
    assert np.array_equal(got_s, exp_s) and got_t == exp_t


def test_data_association_contention(ctx, oracle, pkg):
    """many queries claim the same train row: exercises the Q1 counter/steal logic heavily."""
    from ssk_amd import capi
    rng = np.random.default_rng(99)
    a = synth.synth_kf(rng, 500, 500, 400)
    b = synth.synth_kf(rng, 500, 500, 12)
    src = b["desc_left"][b["matches"]["queryIdx"]]
    for k, i in enumerate(a["matches"]["queryIdx"]):
        a["desc_left"][i] = synth.flip_bits(rng, src[k % len(src)][None], int(rng.integers(0, 40)))[0]
    cp, op = prm_pair(pkg, (0, 1, 370, 60.0))
    got = ctx.data_association(capi.make_kf_view(**a)[0], capi.make_kf_view(**b)[0], cp)
    exp = oracle.data_association(ol.make_kf_view(**a)[0], ol.make_kf_view(**b)[0], op)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[2] == exp[2]
    assert set(np.unique(got[1])) >= {0, 1}


def test_data_association_empty_other(ctx, oracle, pkg):
    from ssk_amd import capi
    rng = np.random.default_rng(3)
    a = synth.synth_kf(rng, 50, 50, 20)
    b = synth.synth_kf(rng, 50, 50, 0)
    cp, op = prm_pair(pkg, (1, 1, 370, 60.0))
    got = ctx.data_association(capi.make_kf_view(**a)[0], capi.make_kf_view(**b)[0], cp)
    exp = oracle.data_association(ol.make_kf_view(**a)[0], ol.make_kf_view(**b)[0], op)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[2] == exp[2] == 0


def test_golden_da_gpu(ctx, pkg):
    import os
    from ssk_amd import capi
    for name in ("da_case0.npz", "da_case1.npz"):
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
        a = {k[2:]: g[k] for k in g.files if k.startswith("a_")}
        b = {k[2:]: g[k] for k in g.files if k.startswith("b_")}
        prm = capi.DaParams(int(g["prm"][0]), int(g["prm"][1]), int(g["prm"][2]), float(g["prm"][3]))
        va, ka = capi.make_kf_view(**a)
        vb, kb = capi.make_kf_view(**b)
        out, st, tracked = ctx.data_association(va, vb, prm)
        assert np.array_equal(out["trainIdx"], g["exp_trainIdx"]) and np.array_equal(out["distance"], g["exp_distance"])
        assert np.array_equal(st, g["exp_status"]) and tracked == int(g["exp_tracked"])


def test_data_association_staged_equals_the_three_calls(ctx):
    """ssk_data_association_staged (one upload, one synchronisation) against ssk_data_association + ssk_da_fundamental_filter +
    ssk_da_change_in_pose on the same keyframes: identical matches, statuses, counts and results, with every combination of
    the two optional stages, from the identity and from a given start pose, and for an empty other keyframe."""
    import ctypes as C
    from ssk_amd import capi
    fp = capi.FrontendParams(**synth.KITTI_FRONTEND)
    frames, _ = synth.synth_translation_sequence(3, 2)          # 0.54 m of motion between the two pairs (a pure image shift is degenerate for F)
    a = ctx.frontend_run(frames[1][0], frames[1][1], fp)
    b = ctx.frontend_run(frames[0][0], frames[0][1], fp)
    va, _ka = capi.make_kf_view(**a)
    vb, _kb = capi.make_kf_view(**b)
    dap = capi.DaParams(1, 1, synth.H, 60.0)
    fmp, cpp = capi.FmParams(), capi.CipParams()
    o = ol.kitti_fm_params(seed=0x5EED0123, norm_cx=0.5 * synth.W, norm_cy=0.5 * synth.H, norm_scale=2.0 / synth.W)
    C.memmove(C.byref(fmp), C.byref(o), C.sizeof(fmp))
    o = ol.kitti_cip_params()
    C.memmove(C.byref(cpp), C.byref(o), C.sizeof(cpp))
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    start = ident.copy(); start[3] = 0.05; start[11] = -0.1
    m0, st0, tr0 = ctx.data_association(va, vb, dap)
    assert tr0 >= 100
    for use_fm in (False, True):
        for use_cip in (False, True):
            for init in (None, start):
                st, trs = st0.copy(), [tr0]
                fr = cr = None
                if use_fm:
                    st, fr = ctx.da_fundamental_filter(va, vb, m0, st, fmp)
                trs.append(int((st == 0).sum()))
                if use_cip:
                    st, _res, cr = ctx.da_change_in_pose(va, vb, m0, st, ident if init is None else init, cpp)
                trs.append(int((st == 0).sum()))
                m1, st1, so = ctx.data_association_staged(va, vb, dap, fmp if use_fm else None, cpp if use_cip else None, init)
                assert np.array_equal(m1, m0) and np.array_equal(st1, st), (use_fm, use_cip)
                assert [so.tracked_da, so.tracked_fm, so.tracked_final] == trs
                if use_fm:
                    assert bytes(so.fm) == bytes(fr) and so.fm.valid == 1
                if use_cip:
                    assert bytes(so.cip) == bytes(cr) and so.cip.valid == 1
    # an other keyframe without matches: nothing tracked, the later stages leave the statuses alone
    e = {k: v[:0] for k, v in b.items()}
    ve, _ke = capi.make_kf_view(**e)
    m2, st2, tr2 = ctx.data_association(va, ve, dap)
    m3, st3, so = ctx.data_association_staged(va, ve, dap, fmp, cpp, None)
    assert tr2 == 0 and np.array_equal(st3, st2) and np.array_equal(m3["trainIdx"], m2["trainIdx"]) and so.tracked_final == 0


def test_data_association_staged_multi_equals_pair_by_pair(ctx):
    """One keyframe against three others (one of them empty) in ONE call = the pairs one at a time: per-pair RANSAC seeds and
    start poses reach their pair, rows do not leak into each other."""
    import ctypes as C
    from ssk_amd import capi
    fp = capi.FrontendParams(**synth.KITTI_FRONTEND)
    frames, _ = synth.synth_translation_sequence(3, 4)
    kfs = [ctx.frontend_run(l, r, fp) for l, r in frames]
    va, _ka = capi.make_kf_view(**kfs[3])
    others = [kfs[2], {k: v[:0] for k, v in kfs[0].items()}, kfs[1]]
    keep = [capi.make_kf_view(**o) for o in others]
    views = [k[0] for k in keep]
    dap = capi.DaParams(1, 1, synth.H, 60.0)
    cpp = capi.CipParams()
    o = ol.kitti_cip_params()
    C.memmove(C.byref(cpp), C.byref(o), C.sizeof(cpp))
    seeds = [0x5EED0000 + (3 << 20) + 2, 77, 0x5EED0000 + (3 << 20) + 1]
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    poses = np.stack([ident, ident, ident]); poses[2, 11] = -0.5
    fmp = capi.FmParams()
    fo = ol.kitti_fm_params(seed=0, norm_cx=0.5 * synth.W, norm_cy=0.5 * synth.H, norm_scale=2.0 / synth.W)
    C.memmove(C.byref(fmp), C.byref(fo), C.sizeof(fmp))
    m, st, outs = ctx.data_association_staged_multi(va, views, dap, fmp, seeds, cpp, poses)
    for p in range(3):
        one = capi.FmParams()
        fo = ol.kitti_fm_params(seed=seeds[p], norm_cx=0.5 * synth.W, norm_cy=0.5 * synth.H, norm_scale=2.0 / synth.W)
        C.memmove(C.byref(one), C.byref(fo), C.sizeof(one))
        m1, st1, o1 = ctx.data_association_staged(va, views[p], dap, one, cpp, poses[p])
        assert np.array_equal(m[p], m1) and np.array_equal(st[p], st1), p
        assert bytes(outs[p]) == bytes(o1), p
    assert outs[0].tracked_final >= 60 and outs[2].tracked_final >= 60 and outs[1].tracked_final == 0
