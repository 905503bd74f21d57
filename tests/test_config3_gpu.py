"""GPU: BASELINE.json configs[3] at its stated size -- a full KITTI-00-length sequence (4 541 stereo frames) streamed one frame at
a time through the estimator's C-ABI (ssk_slam_process_frame = one iteration of performStereoSLAM, EST.cpp:44-937: odometry,
keyframe policy, place recognition on the device, data association, relative BA), then the global pose graph (EST.cpp:945-987).
The reference's loop-closure detector is DBoW2 on the host; here it is the device-side vocabulary tree of ssk_bow.hip.

No KITTI images exist here: the frames are synthetic KITTI-shaped pairs (tests/synth.py), 0.54 m of motion per frame, the
last 160 frames back over the places just seen so that old keyframes come back into view.  There is no oracle for a whole
sequence; the checks are properties of the domain:
  * no frame with invalid odometry, a keyframe every 4th frame (0.54 m per frame against the 2 m limit of KITTI.ini:127),
  * every keyframe solve ends below its initial error,
  * the revisit links the new keyframes to OLD local maps: keyframes with more than one new kf-to-kf edge (case B of
    edge_creation_policy, SRBA.h:242-302) -- the loop closing of this graph; the report's `confirmed_loop_closure` flag is the
    reference's narrower notion (a query result BELOW 0.8 x the best score that is topologically far, EST.cpp:1794-1826) and
    is only counted,
  * drift after the global pose graph.  The graph of this trajectory is a CHAIN of 1 135 keyframes: the revisit at the far end ties the
    last 86 m to themselves, nothing ties the far end to the start, so what is checked is dead reckoning: a random walk whose size depends
    on the frame source's seed.  Round 4 measured the DISTRIBUTION (tools/drift_sweep.py, profiles/r04_drift_sweep.json: five seeds, both
    ORB steering forms): along track 0.008-0.25 %, 3-D 0.15-6.1 % of the path, seed 9 -- the only seed of rounds 1-3 -- the outlier in both
    forms (6.1 % / 3.3 %); the medians are 0.011 % and 0.47 %.  The test runs seeds 9, 10, 13 -- the three of the sweep whose revisit links
    keyframes to old local maps (6, 18 and 1 keyframes with extra edges; seeds 11 and 12 close no loop at all, so they cannot guard the
    loop-closing path) -- and asserts the MEDIAN against the bounds of rounds 1-2 (0.2 % along track, 3.5 % in 3-D, 0.1 rad of heading) and
    every single run against the 3 sigma bounds (0.5 %, 12 %, 0.25 rad),
  * EVERY seed: loop-closing edges beyond the chain exist and the pose-graph optimisation lowers its cost (round 5, ADVICE r4: the checks
    had become conditional on seed 9).
Runs tools/stream_full.py's run(); the same function feeds bench.py's `unbatched` object."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES, N_BACK = 4541, 160


def drift_of(o):
    """(along-track, 3-D, heading) drift after the global pose graph: fractions of the path, largest rotation-matrix deviation"""
    X = o["poses"].reshape(-1, 3, 4)
    kf = np.array(o["kf_frames"])
    truth = np.zeros((len(kf), 3)); truth[:, 1] = -0.54 * np.array(o["kf_pos"])      # robot frame, the camera's right is -y (EST.cpp:1105-1112)
    path = 0.54 * (N_FRAMES - 1)
    err = np.linalg.norm(X[:, :, 3] - truth, axis=1)
    along = np.abs(X[:, 1, 3] - truth[:, 1])
    return along.max() / path, err.max() / path, float(np.abs(X[:, :, :3] - np.eye(3)).max())


def test_config3_full_length_stream(pkg):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stream_full
    drifts = []
    for seed in (9, 10, 13):
        o = stream_full.run(N_FRAMES, N_BACK, seed=seed, verbose=False)
        assert o["frames"] == N_FRAMES and o["invalid_odometry"] == [] and o["error_increased"] == [], seed
        kf = np.array(o["kf_frames"])
        assert o["keyframes"] == len(kf) == 1 + (N_FRAMES - 1) // 4 and np.all(np.diff(kf) == 4)
        assert o["landmarks"] > 100 * 1000
        along, err3, heading = drift_of(o)
        drifts.append((along, err3, heading))
        # every run: the 3 sigma bounds of the random walk (module header)
        assert along < 0.005 and err3 < 0.12 and heading < 0.25, (seed, along, err3, heading)
        path = 0.54 * (N_FRAMES - 1)
        X = o["poses"].reshape(-1, 3, 4)
        # the keyframes of the way back sit on the places of the way out
        assert abs(X[-1, 1, 3] + 0.54 * o["kf_pos"][-1]) < 0.005 * path
        # loop-closing edges beyond the chain on every seed (a regression that loses them must not pass), and a pose graph that has something to move
        assert o["multi_edge_kfs"] >= 1 and o["edges"] >= o["keyframes"] - 1 + 1, (seed, o["multi_edge_kfs"], o["edges"])
        assert o["pg_iters"] >= 1 and o["pg_cost_final"] < o["pg_cost_init"], (seed, o["pg_cost_init"], o["pg_cost_final"])
        t = o["timing"]
        assert t["frames"] == N_FRAMES and t["kf_frames"] == o["keyframes"] - 1 and t["lm_solves"] == 2 * t["kf_frames"]       # stage 1 + stage 2 per keyframe
        if seed == 9:
            assert o["pg_cost_final"] < 0.1 * o["pg_cost_init"]
        print("configs[3] seed %d: %d frames, %d keyframes, %d edges, %.1f frames/s un-batched, %.2f ms per keyframe frame; confirmed_loop_closure flags %d, "
              "keyframes linked to an old local map %d; drift along track %.3f %%, 3-D %.2f %%, heading %.3f; pose graph %.4g -> %.4g"
              % (seed, N_FRAMES, o["keyframes"], o["edges"], o["frames_per_s"], o["ms_per_keyframe_frame"], len(o["loop_closures"]), o["multi_edge_kfs"],
                 100 * along, 100 * err3, heading, o["pg_cost_init"], o["pg_cost_final"]))
    # the distribution: medians against the bounds of rounds 1-2 (a regression of the LM, the data association or the facade moves every seed)
    med = np.median(np.array(drifts), axis=0)
    assert med[0] < 0.002 and med[1] < 0.035 and med[2] < 0.1, med


def test_revisit_of_the_start_closes_the_loop(pkg):
    """Out and ALL the way back (1 201 frames: 600 out, 600 back over the same places): the keyframes of the way back are linked to the
    old local maps (case B of edge_creation_policy, SRBA.h:242-302), so the last keyframe -- back at the start -- hangs on keyframe 0's
    neighbourhood by a SHORT chain of edges instead of the 300-edge way round: its position error is that of a few edges, not the
    accumulated drift of the far end, and the pose graph has loop constraints to lower its cost with."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stream_full
    n, n_back = 1201, 600
    o = stream_full.run(n, n_back, verbose=False)
    assert o["invalid_odometry"] == [] and o["error_increased"] == []
    assert o["multi_edge_kfs"] >= 3                                                       # the way back is tied to the way out again and again
    X = o["poses"].reshape(-1, 3, 4)
    kp = np.array(o["kf_pos"])
    truth = np.zeros((len(kp), 3)); truth[:, 1] = -0.54 * kp
    err = np.linalg.norm(X[:, :, 3] - truth, axis=1)
    path = 0.54 * (n - 1)
    assert kp[-1] <= 4 and err[-1] < 0.005 * path, (kp[-1], err[-1], path)                # back at the start: within 0.5 % of the 648 m driven
    assert err.max() < 0.05 * (0.54 * n_back)                                             # the far end: dead reckoning over half the path at most
    assert o["pg_cost_final"] <= o["pg_cost_init"]
    print("revisit of the start: %d keyframes, %d linked to an old local map, error of the last keyframe (back at the start) %.3f m, largest %.3f m over a %.0f m path; pose graph %.4g -> %.4g"
          % (o["keyframes"], o["multi_edge_kfs"], err[-1], err.max(), path, o["pg_cost_init"], o["pg_cost_final"]))


def test_literal_loop_closing_edge_init_diverges(pkg):
    """Why `literal_lc_edge_init` is NOT the default: the same 4 541-frame sequence with SRBA.h:294 as written -- an
    extra edge from a new local-map centre to a FAR centre starts from a copy of the edge created just before, i.e. off by the
    distance between the two maps.  The topology is identical (tests/test_edge_policy_gpu.py compares it with the literal oracle);
    the robust LM of this build does not pull such an edge back: the keyframes of the way back end hundreds of metres from their
    places and the pose graph starts from a cost four orders of magnitude higher.  With the default (initial value composed from the
    current estimates along the graph) the same run is test_config3_full_length_stream above."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stream_full
    o = stream_full.run(N_FRAMES, N_BACK, verbose=False, literal_lc_edge_init=1.0)
    assert o["invalid_odometry"] == [] and o["multi_edge_kfs"] >= 1          # the same revisit, the same extra edges
    X = o["poses"].reshape(-1, 3, 4)
    along = np.abs(X[:, 1, 3] + 0.54 * np.array(o["kf_pos"]))
    path = 0.54 * (N_FRAMES - 1)
    assert along.max() > 0.1 * path and o["pg_cost_init"] > 1e3, (along.max(), o["pg_cost_init"])
