"""GPU: BASELINE.json configs[0] at its stated size -- the reference's demo parameter file (demo/config_imgdir_kitti_srba.ini, the
"kitti" preset pinned to it by tests/test_kitti_ini.py) on the first 50 stereo pairs of a KITTI-shaped sequence, through the
estimator's C-ABI (include/ssk_slam.h: ssk_slam_process_frame = one iteration of performStereoSLAM, EST.cpp:44-937).

No KITTI images exist here, so the pairs are synthetic (1241x376, 0.54 m of motion per frame: SURVEY 8(d) config 1).  For EVERY
keyframe the loop inserts, the oracle re-does, from the same images and the same inputs, what the device did:
  * the stereo front-end (keypoints, descriptors, L<->R matches: every field, bit for bit),
  * the data association with every similar keyframe (Hamming K=1, direction and ORB-distance filters, fundamental-matrix
    RANSAC, change-in-pose filter from the same initial pose): tracking_info entry by entry,
  * the landmark-id propagation (EST.cpp:572-617),
and the debug dumps the reference writes per keyframe (info_kf%04d.txt KF.cpp:60-91, da_info_%04d.txt EST.cpp:751-763) are
parsed and compared with the keyframe the estimator holds."""
import os

import numpy as np
import pytest

import bow_vocab as bv
import oracle_lib as ol
import synth

pytestmark = pytest.mark.gpu
N_PAIRS = 50


def stereo_obs(kf):
    m = kf["matches"]
    return np.stack([kf["kp_left"]["x"][m["queryIdx"]], kf["kp_left"]["y"][m["queryIdx"]],
                     kf["kp_right"]["x"][m["trainIdx"]], kf["kp_right"]["y"][m["trainIdx"]]], 1)


def oracle_da(oracle, this_kf, other_kf, this_id, other_id, init_pose):
    """m_internal_data_association (EST.cpp:1413-1669) with the KITTI parameters, stage by stage on the CPU."""
    va, _ka = ol.make_kf_view(**this_kf)
    vb, _kb = ol.make_kf_view(**other_kf)
    m, st, tracked = oracle.data_association(va, vb, ol.DaParams(1, 1, synth.H, 60.0))
    n = len(m)
    first = -np.ones(n, np.int32)
    if tracked < 15:                                          # EST.cpp:1590-1594
        return first, np.zeros(n), 0
    so_a, so_b = stereo_obs(this_kf), stereo_obs(other_kf)
    fmp = ol.kitti_fm_params(seed=0x5EED0000 + (this_id << 20) + other_id, norm_cx=0.5 * synth.W, norm_cy=0.5 * synth.H, norm_scale=2.0 / synth.W)
    st, _r = oracle.fundamental_filter(so_a[:, :2], so_b[:, :2], m["trainIdx"], st, fmp)
    if int((st == 0).sum()) < 15:                             # EST.cpp:1617-1621
        return first, np.zeros(n), 0
    st, res, out = oracle.change_in_pose(so_a, so_b, m["trainIdx"], st, init_pose, ol.kitti_cip_params())
    ok = st == 0
    first[ok] = m["trainIdx"][ok]
    return first, np.where(ok, m["distance"].astype(np.float64), 0.0), int(ok.sum())


def test_config0_fifty_pairs_every_keyframe_against_oracle(pkg, ctx, oracle, tmp_path):
    from ssk_amd import capi
    frames, _ = synth.synth_translation_sequence(3, N_PAIRS)
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    oprm = ol.FrontendParams(**synth.KITTI_FRONTEND)
    pat = ol.default_pattern()
    voc = os.path.join(tmp_path, "voc.yml.gz")
    bv.write_yaml(bv.train([ctx.frontend_run(l, r, prm)["desc_left"] for l, r in frames[::5]], k=8, L=3, seed=5), voc)
    slam = capi.Slam(ctx, synth.W, synth.H, "kitti")
    slam.set_debug_dir(tmp_path)
    slam.load_vocabulary(voc)
    kfs, kf_ids, last_id, n_da_pairs = [], [], 0, 0
    for k, (l, r) in enumerate(frames):
        rep = slam.process_frame(l, r)
        assert rep.vo_valid == 1, k
        if not rep.inserted_kf:
            continue
        kid = rep.kf_id if k > 0 else 0
        assert kid == len(kfs)
        got, ids = slam.get_keyframe(kid)
        # ---- front-end: every field of every keypoint / descriptor / match (criterion of UTIL.cpp:33-80) ----
        exp = oracle.frontend(l, r, oprm, pat)
        for name in ("kp_left", "kp_right", "desc_left", "desc_right", "matches"):
            assert np.array_equal(got[name], exp[name]), (k, name)
        assert len(exp["matches"]) >= 120                     # the adaptive loop (EST.cpp:271-315) never left detect_fast_th = 20
        n = len(exp["matches"])
        if kid == 0:
            assert np.array_equal(ids, np.arange(n, dtype=np.uint64))            # generateMatchesIDs(0), KF.h:75-80
            last_id = int(ids[-1])                                              # EST.cpp:115
        else:
            # ---- data association with every similar keyframe, then the id propagation ----
            assert rep.n_similar >= 1 and rep.similar_kfs[0] == kid - 1         # the previous keyframe is always similar (EST.cpp:1758)
            firsts, others, order = [], [], []
            for which in range(rep.n_similar):
                da = slam.last_da(which)
                other = da["other_kf"]
                assert other == rep.similar_kfs[which] and len(da["tracking_first"]) == n
                f, d, tr = oracle_da(oracle, exp, kfs[other], kid, other, da["init_pose"])
                assert tr == da["tracked"], (k, which, tr, da["tracked"])
                assert np.array_equal(f, da["tracking_first"]) and np.array_equal(d, da["tracking_dist"]), (k, which)
                firsts.append(f); others.append(kf_ids[other]); order.append(da["sorted_pos"])
                n_da_pairs += 1
            assert sorted(order) == list(range(rep.n_similar))
            by_pos = np.argsort(order)                                          # ids are taken in DATrackedSorter order (EST.cpp:455-470)
            tracked_counts = [int((firsts[i] >= 0).sum()) for i in by_pos]
            assert tracked_counts == sorted(tracked_counts, reverse=True)
            exp_ids, _per, n_tracked, last_id = oracle.propagate_ids(np.stack([firsts[i] for i in by_pos]), [others[i] for i in by_pos], last_id)
            assert np.array_equal(ids, exp_ids), k
            assert n_tracked >= 60                                              # updated_matches_th (KITTI.ini:128): why it was inserted
        kfs.append(exp); kf_ids.append(ids)
        # ---- the reference's per-keyframe debug dumps ----
        dump = np.loadtxt(os.path.join(tmp_path, "info_kf%04d.txt" % kid), comments="%", ndmin=2)
        assert dump.shape == (n, 7) and np.all(dump[:, 0] == kid) and np.array_equal(dump[:, 1].astype(np.uint64), ids)
        so = stereo_obs(exp)
        assert np.array_equal(dump[:, 2:6], so) and np.array_equal(dump[:, 6], exp["matches"]["distance"])
        head = open(os.path.join(tmp_path, "info_kf%04d.txt" % kid)).readline()
        assert head.startswith("% [KF_ID] [MATCH_ID] [LEFT_PT{x y}] [RIGHT_PT{x y}] [MATCH_DISTANCE]")
        feats = open(os.path.join(tmp_path, "info_feats_kf%04d.txt" % kid)).read().split("\n")
        assert feats[0].split() == [str(len(exp["kp_left"])), str(len(exp["kp_right"]))] and len(feats) == 2 + len(exp["kp_left"]) + len(exp["kp_right"])
        if kid > 0:
            dai = np.loadtxt(os.path.join(tmp_path, "da_info_%04d.txt" % kid), ndmin=2)          # the exact SRBA input: id + 4 pixels
            assert dai.shape == (n, 5) and np.array_equal(dai[:, 0].astype(np.uint64), ids) and np.array_equal(dai[:, 1:], so)
            assert rep.n_created_edges >= 1 and rep.total_sqr_error_final <= rep.total_sqr_error_init
    # 0.54 m per frame against the 2 m limit: a keyframe every 4th frame (EST.cpp:366-392)
    nk, ne, nl = slam.counts()
    assert nk == len(kfs) == 1 + (N_PAIRS - 1) // 4 and ne >= nk - 1 and nl > 500 and n_da_pairs >= nk - 1
    slam.close()
