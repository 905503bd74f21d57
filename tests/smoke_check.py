"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0 -- front-end (seam B1), data association
(seam B2) and one LM solve (seam B3) -- checked against the oracle (the checker, never the thing shipped)."""
import ctypes as C

import numpy as np


def run():
    import __graft_entry__ as ge
    import oracle_lib as ol
    import synth
    import synth_lm as sl
    pkg = ge.load_package()
    from ssk_amd import capi
    oracle = ol.load()
    ctx = pkg.Context(0)

    # B1: one 1241x376 stereo pair through the HIP front-end, every field bit-exact
    left, right = synth.synth_stereo_pair(1)
    got = ctx.frontend_run(left, right, capi.FrontendParams(**synth.KITTI_FRONTEND))
    exp = oracle.frontend(left, right, ol.FrontendParams(**synth.KITTI_FRONTEND), ol.default_pattern())
    for k in ("kp_left", "kp_right", "matches"):
        assert len(got[k]) == len(exp[k]), k
        for f in got[k].dtype.names:
            assert np.array_equal(got[k][f], exp[k][f]), (k, f)
    assert np.array_equal(got["desc_left"], exp["desc_left"]) and np.array_equal(got["desc_right"], exp["desc_right"])
    print("smoke: front-end on cuda:0 matches the oracle bit-exactly (%d + %d keypoints, %d stereo matches)"
          % (len(got["kp_left"]), len(got["kp_right"]), len(got["matches"])))

    # B2: Hamming K=1 + direction / ORB-distance filters
    rng = np.random.default_rng(2026)
    a, b = synth.synth_kf_pair(rng, 500, 300)
    vals = (1, 1, 370, 60.0)
    got = ctx.data_association(capi.make_kf_view(**a)[0], capi.make_kf_view(**b)[0], capi.DaParams(*vals))
    exp = oracle.data_association(ol.make_kf_view(**a)[0], ol.make_kf_view(**b)[0], ol.DaParams(*vals))
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and got[2] == exp[2]
    print("smoke: data association on cuda:0 matches the oracle bit-exactly (%d tracked of %d)" % (got[2], len(got[1])))

    # B3: one relative-BA window, final error and state within the north-star tolerance
    win, _ = sl.make_submap_problem(1, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    prm = sl.make_params()
    ref = oracle.lm_solve(win, prm)
    cp = capi.LmParams()
    C.memmove(C.byref(cp), C.byref(prm), C.sizeof(cp))
    batch = capi.LmBatch(ctx, [win])
    batch.run(cp)
    res = batch.results()[0]
    poses, lms = batch.state(0)
    assert res.iters == ref["res"].iters and abs(res.err_final - ref["res"].err_final) <= 1e-6 * ref["res"].err_final
    assert np.abs(poses - ref["poses"]).max() <= 1e-5 and np.abs(lms - ref["lms"]).max() <= 1e-4
    print("smoke: LM solve on cuda:0 matches the oracle (%d iterations, error %.6g -> %.6g)" % (res.iters, res.err_init, res.err_final))
    batch.close()

    # place recognition: vocabulary-tree transform + L1 database query, bit for bit
    import bow_vocab as bv
    v = bv.synth(k=8, L=3, seed=3, ragged=True)
    G, O = capi.BowVocab(ctx, v), oracle.bow_vocab(v)
    gdb, odb = capi.BowDb(ctx, G, 16, 256), oracle.bow_db(O.n_words)
    seq = bv.keyframe_sequence(v, np.random.default_rng(4), n_kf=8, n=200)
    for f in seq:
        gw, gx = G.transform(f)
        ow, ox = O.transform(f)
        assert np.array_equal(gw, ow) and np.array_equal(gx, ox)
        gi, gs = gdb.query(gw, gx, 4)
        oi, os_ = odb.query(ow, ox, 4)
        assert np.array_equal(gi, oi) and np.array_equal(gs, os_)
        gdb.add(gw, gx); odb.add(ow, ox)
    print("smoke: bag-of-words transform + database query on cuda:0 match the oracle bit-exactly (%d words, best score %.4f)" % (len(gw), gs[0]))
    gdb.close(); G.close()
    ctx.close()


if __name__ == "__main__":
    run()
