"""GPU box (the engine needs a context; with run_local_optimization = 0 no kernel runs): mySRBA::edge_creation_policy, SURVEY 8(a) row
a13.  The facade's restatement (host/srba_stereo_slam.cpp, driven through the C-ABI ssk_rba_define_kf) against the literal oracle
oracle/oracle_edge_policy.c (SRBA.h:144-308) on the same observation streams: for every keyframe the created kf-to-kf edges
(from, to), the local-map sets (m_kf_localmap_center_ID, m_localmap_kf_IDs) and the INITIAL inv_pose of every new edge.
Streams: a plain chain through several local maps, and a revisit whose new local-map centre (case B, SRBA.h:216-306) sees landmarks
of its own previous map AND of two far, old maps -- three edges from one keyframe, the second and third to far centres."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def obs_of(ids, rng):
    from ssk_amd import capi
    o = np.zeros(len(ids), capi.RBA_OBS_DTYPE)
    o["feat_id"] = ids
    o["left_x"] = rng.uniform(100, 1100, len(ids)); o["left_y"] = rng.uniform(50, 320, len(ids))
    o["right_x"] = o["left_x"] - rng.uniform(5, 60, len(ids)); o["right_y"] = o["left_y"]
    o["rel_pos"] = np.column_stack([rng.uniform(4, 30, len(ids)), rng.uniform(-5, 5, len(ids)), rng.uniform(-1, 3, len(ids))])
    return o


def small_pose(rng):
    a = rng.normal(0, 0.03)
    return np.array([np.cos(a), -np.sin(a), 0, rng.normal(1.0, 0.1), np.sin(a), np.cos(a), 0, rng.normal(0, 0.05), 0, 0, 1, 0.0])


def drive(ctx, oracle, streams, literal, submap_size=4, max_tree_depth=3, max_optimize_depth=2, min_obs=6):
    from ssk_amd import capi
    rng = np.random.default_rng(7)
    rba = capi.Rba(ctx, capi.RbaOptions.kitti(submap_size=submap_size, max_tree_depth=max_tree_depth, max_optimize_depth=max_optimize_depth,
                                              min_obs_to_loop_closure=min_obs, reserved=1 if literal else 0))
    ecp = ol.OracleEdgePolicy(oracle.lib, submap_size, min_obs, max_tree_depth, max_optimize_depth)
    created = []
    for kf, ids in enumerate(streams):
        init = small_pose(rng)
        rba.set_initial_kf_pose(init); ecp.set_initial_kf_pose(init)
        info = rba.define_kf(obs_of(ids, rng), optimize=False)
        n = ecp.define_kf(ids)
        assert info.kf_id == kf and info.n_created_edges == n, (kf, info.n_created_edges, n)
        assert rba.counts()[1] == ecp.num_edges()
        new = []
        for k in range(n):
            e = info.created_edge_ids[k]
            assert e == ecp.num_edges() - n + k
            fg, tg, pg = rba.edge(e)
            fo, to, po = ecp.edge(e)
            assert (fg, tg) == (fo, to), (kf, e)
            new.append((fg, tg, pg, po))
        created.append(new)
        for q in range(kf + 1):
            if kf >= 1:
                assert rba.localmap_centers(q) == ecp.kf_centers(q), (kf, q)
            assert rba.localmap_kfs(q) == ecp.localmap_kfs(q), (kf, q)
    rba.close()
    return created


def chain_stream(n_kf, per_kf=40, step=12):
    """keyframe k sees ids [step k, step k + per_kf): consecutive keyframes share most of them"""
    return [np.arange(step * k, step * k + per_kf, dtype=np.uint64) for k in range(n_kf)]


def test_chain_of_local_maps_matches_the_literal_policy(ctx, oracle):
    created = drive(ctx, oracle, chain_stream(14), literal=True)
    assert [len(c) for c in created] == [0] + [1] * 13
    for new in created:
        for f, t, pg, po in new:
            assert np.array_equal(pg, po)                       # initial inv_pose: SRBA.h:194 / :203, bit for bit
    # submap_size 4: keyframe 0 centres {1, 2, 3}, keyframe 4 is the next centre, linked to centre 0 (SRBA.h:176-209, 216-306)
    assert [(f, t) for new in created for f, t, _, _ in new][:5] == [(0, 1), (0, 2), (0, 3), (0, 4), (4, 5)]


def test_case_b_keyframe_linking_far_centres(ctx, oracle):
    """16 keyframes of a chain (centres 0, 4, 8, 12), then keyframe 16 -- a new centre -- that re-observes landmarks based in the maps
    of centres 0 and 4 besides its own neighbourhood: edges to 12 (near: its own previous map), then to the far centres in the
    order of their votes."""
    streams = chain_stream(16)
    own = np.arange(12 * 15, 12 * 15 + 40, dtype=np.uint64)            # what keyframe 15 saw
    old0 = np.arange(0, 30, dtype=np.uint64)                           # based at keyframes 0..2 (map of centre 0)
    old4 = np.arange(12 * 5 + 28, 12 * 5 + 40, dtype=np.uint64)        # first seen at keyframe 5..: map of centre 4
    streams.append(np.concatenate([own, old0, old4]))
    for literal in (True, False):
        created = drive(ctx, oracle, streams, literal=literal)
        last = created[16]
        assert len(last) == 3, [(f, t) for f, t, _, _ in last]
        assert {(f, t) for f, t, _, _ in last} == {(12, 16), (0, 16), (4, 16)}
        assert last[0][0] == 12 and last[0][1] == 16           # most votes: its own previous map
        f0, t0, pg0, po0 = last[0]
        assert np.array_equal(pg0, po0)                         # the first edge: SRBA.h:294 = a copy of the previous edge's value, both ways
        for f, t, pg, po in last[1:]:
            if literal:
                assert np.array_equal(pg, po)                   # as written: every further edge copies the edge created just before
            else:
                # the build's default: from the composed current estimates along the graph -- NOT the literal value (documented
                # deviation); the topology is the literal one either way
                assert not np.array_equal(pg, po)
