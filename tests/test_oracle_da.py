"""CPU tests of the oracle's data-association restatement (known answers + golden fixture).
The reference holds no tests or vectors for this path: these known-answer cases are
derived by hand from the cited source lines."""
import os

import numpy as np
import pytest

import oracle_lib as ol
import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def dm(pairs):
    m = np.zeros(len(pairs), ol.DMATCH_DTYPE)
    for k, (t, d) in enumerate(pairs):
        m[k] = (k, t, 0, d)
    return m


def test_hamming_known_answers(oracle):
    z = np.zeros((1, 32), np.uint8)
    o = np.full((1, 32), 255, np.uint8)
    r = oracle.hamming_match(z, np.vstack([o, z]))
    assert r["trainIdx"][0] == 1 and r["distance"][0] == 0.0 and r["imgIdx"][0] == 0 and r["queryIdx"][0] == 0
    assert oracle.hamming_match(z, o)["distance"][0] == 256.0
    # tie -> first (lowest) train index, OpenCV's strict '<' scan
    t = np.zeros((3, 32), np.uint8)
    t[0, 0] = 0x03
    t[1, 5] = 0x81
    t[2, 31] = 0x01
    q = np.zeros((1, 32), np.uint8)
    q[0, 31] = 0x03   # distances: 4, 4, 1 -> idx 2
    assert oracle.hamming_match(q, t)["trainIdx"][0] == 2
    q[0, 31] = 0x00   # distances 2,2,1 -> 2 ; remove row 2 -> tie 2,2 -> 0
    assert oracle.hamming_match(q, t[:2])["trainIdx"][0] == 0
    # empty train or query -> BFMatcher clears the output
    assert len(oracle.hamming_match(q, np.zeros((0, 32), np.uint8))) == 0
    assert len(oracle.hamming_match(np.zeros((0, 32), np.uint8), t)) == 0


def test_hamming_matches_numpy(oracle):
    rng = np.random.default_rng(1)
    q = synth.random_descriptors(rng, 77)
    t = synth.random_descriptors(rng, 130)
    t[40] = q[3]
    t[90] = q[3]          # exact duplicate rows: tie at distance 0 -> 40
    r = oracle.hamming_match(q, t)
    d = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=2).sum(axis=2)
    assert np.array_equal(r["trainIdx"], d.argmin(axis=1))     # numpy argmin = first minimum
    assert np.array_equal(r["distance"], d.min(axis=1).astype(np.float32))
    assert r["trainIdx"][3] == 40


def test_direction_bins_q2(oracle):
    """EST.cpp:1915: ang = 180/pi*atanf(dy/dx)+180 lies in [90,270]; dx==0 -> +inf -> bin 27."""
    # 6 matches pointing the same way (45 deg -> ang 225 -> bin 22), one at dx=0 (bin 27), one at -45 (bin 13)
    this_xy = np.zeros((8, 2), np.float32)
    other_xy = np.array([[10, 10]] * 6 + [[0, 7], [10, -10]], np.float32)
    m = dm([(k, 5) for k in range(8)])
    st = oracle.da_filter_direction(m, this_xy, other_xy, 0, np.zeros(8, np.int32))
    assert st.tolist() == [0] * 6 + [2, 2]
    # neighbours of the mode survive: 56 deg -> ang 236 -> bin 23 ; 34 deg -> ang 214 -> bin 21 ; 20 deg -> bin 20 rejected
    other_xy = np.array([[10, 10]] * 4 + [[10, 14.8], [10, 6.75], [10, 3.64]], np.float32)
    m = dm([(k, 5) for k in range(7)])
    st = oracle.da_filter_direction(m, this_xy[:7], other_xy, 0, np.zeros(7, np.int32))
    assert st.tolist() == [0, 0, 0, 0, 0, 0, 2]
    # offset is added to the OTHER keypoint's row (EST.cpp:1913)
    st2 = oracle.da_filter_direction(dm([(0, 5), (1, 5), (2, 5)]), np.zeros((3, 2), np.float32),
                                     np.array([[100, 0], [100, 0], [-100, 0]], np.float32), 370, np.zeros(3, np.int32))
    assert st2.tolist() == [0, 0, 2]
    # first maximal bin wins on a tie (strict '>', :1926): two bins with 2 each -> lower bin is the mode
    other_xy = np.array([[10, -10], [10, -10], [10, 10], [10, 10]], np.float32)   # bins 13,13,22,22
    st3 = oracle.da_filter_direction(dm([(k, 5) for k in range(4)]), np.zeros((4, 2), np.float32), other_xy, 0,
                                     np.zeros(4, np.int32))
    assert st3.tolist() == [0, 0, 2, 2]
    # already-rejected matches are not histogrammed and keep their status (:1903)
    st4 = oracle.da_filter_direction(dm([(k, 5) for k in range(4)]), np.zeros((4, 2), np.float32), other_xy, 0,
                                     np.array([3, 3, 0, 0], np.int32))
    assert st4.tolist() == [3, 3, 0, 0]


def test_orb_distance_quirk_q1(oracle):
    """EST.cpp:1951-2010 traced by hand; other_matched[t].second holds ++stage1_counter (Q1)."""
    # k0: t=0 d=10 -> accepted, counter=1, second[0]=1
    # k1: t=0 d=5  -> 5 > 1 -> NON_TRACKED   (a *better* match loses: the quirk)
    # k2: t=1 d=61 -> REJ_ORB
    # k3: t=1 d=0  -> 0 > FLT_MAX no; accepted, counter=2, second[1]=2
    # k4: t=1 d=2  -> 2 > 2 no -> steals: status[3]=NON_TRACKED, counter 2->1->2, second[1]=2
    # k5: t=0 d=1  -> 1 > 1 no -> steals from k0: status[0]=NON_TRACKED, counter ->1->2, second[0]=2
    m = dm([(0, 10), (0, 5), (1, 61), (1, 0), (1, 2), (0, 1)])
    st = oracle.da_filter_orb_distance(m, 2, 60.0, np.zeros(6, np.int32))
    assert st.tolist() == [1, 1, 3, 1, 0, 0]
    # threshold is strict '>' (:1980): distance == max is kept
    st = oracle.da_filter_orb_distance(dm([(0, 60), (1, 61)]), 2, 60.0, np.zeros(2, np.int32))
    assert st.tolist() == [0, 3]
    # previously rejected matches are skipped (:1969)
    st = oracle.da_filter_orb_distance(dm([(0, 10), (0, 0)]), 1, 60.0, np.array([2, 0], np.int32))
    assert st.tolist() == [2, 0]


def test_propagate_ids(oracle):
    """EST.cpp:572-617: first similar KF (given order) that tracks the match donates its ID; an ID already
    used in this KF makes the match a new feature (the 'break')."""
    tf = np.array([[0, 1, -1, 1, -1],
                   [2, -1, 0, 2, -1]], np.int32)
    ids_a = np.array([100, 101, 102], np.uint64)
    ids_b = np.array([200, 201, 100], np.uint64)
    out, per, n_tracked, last = oracle.propagate_ids(tf, [ids_a, ids_b], 500)
    # m0: kfA idx0 -> 100 ; m1: kfA idx1 -> 101 ; m2: kfB idx0 -> 200 ; m3: kfA idx1 -> 101 dup -> break -> new 501
    # m4: untracked -> 502
    assert out.tolist() == [100, 101, 200, 501, 502]
    assert per.tolist() == [2, 1] and n_tracked == 3 and last == 502


def test_project_match_to_3d(oracle):
    k = synth.KITTI
    cam = ol.StereoCam(k["fx"], k["fy"], k["cx"], k["cy"], k["fx"], k["fy"], k["cx"], k["cy"], k["baseline"])
    # closed form for identical intrinsics: Z = f*b/d, X = (u-cx)*Z/f, Y = (v-cy)*Z/f
    ul, vl, ur = 700.0, 150.0, 680.0
    p = oracle.project_match_to_3d(ul, vl, ur, cam)
    z = k["fx"] * k["baseline"] / (ul - ur)
    assert np.allclose(p, [(ul - k["cx"]) * z / k["fx"], (vl - k["cy"]) * z / k["fx"], z], rtol=1e-14)
    # literal formula (UTIL.h:572-573), bit for bit
    b_d = k["baseline"] / (k["fx"] * (k["cx"] - ur) + k["fx"] * (ul - k["cx"]))
    assert p.tolist() == [b_d * k["fx"] * (ul - k["cx"]), b_d * k["fx"] * (vl - k["cy"]), b_d * k["fx"] * k["fx"]]


def test_data_association_gather_q3(oracle):
    """Q3: DA runs on matched-only descriptors re-indexed by match (EST.cpp:1354-1358,1435-1439)."""
    rng = np.random.default_rng(7)
    a, b = synth.synth_kf_pair(rng, 120, 60)
    va, ka = ol.make_kf_view(**a)
    vb, kb = ol.make_kf_view(**b)
    prm = ol.DaParams(1, 1, 370, 60.0)
    out, st, tracked = oracle.data_association(va, vb, prm)
    curL = a["desc_left"][a["matches"]["queryIdx"]]
    preL = b["desc_left"][b["matches"]["queryIdx"]]
    ref = oracle.hamming_match(curL, preL)
    assert np.array_equal(out, ref)
    axy = np.stack([a["kp_left"]["x"], a["kp_left"]["y"]], 1)[a["matches"]["queryIdx"]]
    bxy = np.stack([b["kp_left"]["x"], b["kp_left"]["y"]], 1)[b["matches"]["queryIdx"]]
    st2, tr2 = oracle.da_filters(ref, len(preL), axy, bxy, prm)
    assert np.array_equal(st, st2) and tracked == tr2 and tracked > 0


@pytest.mark.parametrize("name", ["da_case0.npz", "da_case1.npz"])
def test_golden_da(oracle, name):
    g = np.load(os.path.join(GOLD, name))
    a = {k[2:]: g[k] for k in g.files if k.startswith("a_")}
    b = {k[2:]: g[k] for k in g.files if k.startswith("b_")}
    va, ka = ol.make_kf_view(**a)
    vb, kb = ol.make_kf_view(**b)
    prm = ol.DaParams(int(g["prm"][0]), int(g["prm"][1]), int(g["prm"][2]), float(g["prm"][3]))
    out, st, tracked = oracle.data_association(va, vb, prm)
    assert np.array_equal(out["trainIdx"], g["exp_trainIdx"])
    assert np.array_equal(out["distance"], g["exp_distance"])
    assert np.array_equal(st, g["exp_status"]) and tracked == int(g["exp_tracked"])
