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
    last 86 m to themselves, nothing ties the far end to the start, so what is checked is dead reckoning.  Per edge (2.16 m, ~100
    landmarks at integer keypoint coordinates: a disparity of ~20 px known to +-0.5 px) the along-track error is a few millimetres and
    the heading error ~2 mrad; summed as a random walk over 1 135 edges: ~0.1-0.3 % of the path along the track (two builds whose
    descriptors differ in one bit per hundred keypoints measured 0.09 % and 0.25 %: the bound is 0.5 %), and sqrt(1135) x 2 mrad = 0.07 rad
    of heading (1 sigma), whose integral along the line is ~4 % of the path sideways (the same two builds: 2.8 % and 6.1 %; the bounds are
    3 sigma: 0.25 rad, 12 %).  That a revisit of the START does pull the ends together is the next test,
  * the pose-graph optimisation lowers its cost.
Runs tools/stream_full.py's run(); the same function feeds bench.py's `unbatched` object."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES, N_BACK = 4541, 160


def test_config3_full_length_stream(pkg):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import stream_full
    o = stream_full.run(N_FRAMES, N_BACK, verbose=False)
    assert o["frames"] == N_FRAMES and o["invalid_odometry"] == [] and o["error_increased"] == []
    kf = np.array(o["kf_frames"])
    assert o["keyframes"] == len(kf) == 1 + (N_FRAMES - 1) // 4 and np.all(np.diff(kf) == 4)
    assert o["edges"] >= o["keyframes"] - 1 + 1 and o["multi_edge_kfs"] >= 1          # loop-closing edges beyond the chain
    assert o["landmarks"] > 100 * 1000
    # drift after the global pose graph: robot frame, the camera's right is -y (camera_pose_on_robot, EST.cpp:1105-1112)
    X = o["poses"].reshape(-1, 3, 4)
    truth = np.zeros((len(kf), 3)); truth[:, 1] = -0.54 * np.array(o["kf_pos"])
    err = np.linalg.norm(X[:, :, 3] - truth, axis=1)
    path = 0.54 * (N_FRAMES - 1)
    along = np.abs(X[:, 1, 3] - truth[:, 1])
    assert along.max() < 0.005 * path, (along.max(), path)
    assert err.max() < 0.12 * path, (err.max(), path)
    assert np.abs(X[:, :, :3] - np.eye(3)).max() < 0.25                                   # pure translation: the heading only drifts
    # the keyframes of the way back sit on the places of the way out
    assert abs(X[-1, 1, 3] + 0.54 * o["kf_pos"][-1]) < 0.005 * path
    assert o["pg_iters"] >= 1 and o["pg_cost_final"] < o["pg_cost_init"] and o["pg_cost_final"] < 0.1 * o["pg_cost_init"]
    t = o["timing"]
    assert t["frames"] == N_FRAMES and t["kf_frames"] == o["keyframes"] - 1 and t["lm_solves"] == 2 * t["kf_frames"]       # stage 1 + stage 2 per keyframe
    print("configs[3]: %d frames, %d keyframes, %d edges, %.1f frames/s un-batched, %.2f ms per keyframe frame; confirmed_loop_closure flags %d, "
          "keyframes linked to an old local map %d; max position error %.3f m (along track %.3f m) over %.0f m; pose graph %.4g -> %.4g"
          % (N_FRAMES, o["keyframes"], o["edges"], o["frames_per_s"], o["ms_per_keyframe_frame"], len(o["loop_closures"]), o["multi_edge_kfs"],
             err.max(), along.max(), path, o["pg_cost_init"], o["pg_cost_final"]))


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
