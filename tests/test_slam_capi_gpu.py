"""GPU: the estimator's C-ABI (include/ssk_slam.h): the reference's frame loop one call per stereo pair, from Python."""
import os

import numpy as np
import pytest

import bow_vocab as bv
import synth

pytestmark = pytest.mark.gpu


def test_slam_abi_streams_a_sequence(pkg, ctx, tmp_path):
    from ssk_amd import capi
    frames, _ = synth.synth_translation_sequence(6, 13)
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    voc = os.path.join(tmp_path, "voc.yml.gz")
    bv.write_yaml(bv.train([ctx.frontend_run(l, r, prm)["desc_left"] for l, r in frames[::3]], k=8, L=3, seed=5), voc)
    slam = capi.Slam(ctx, synth.W, synth.H, "kitti")
    with pytest.raises(capi.SskError, match="unknown key"):
        capi.Slam(ctx, synth.W, synth.H, "kitti", no_such_option=1)
    with pytest.raises(capi.SskError):
        slam.load_vocabulary(os.path.join(tmp_path, "missing.yml"))
    slam.load_vocabulary(voc)
    reps = [slam.process_frame(l, r) for l, r in frames]
    # KITTI limits (2 m / 25 deg): a keyframe is forced every 4th frame (4 x 0.54 m > 2 m), as in test_stream_gpu
    assert [k for k, r in enumerate(reps) if r.inserted_kf] == [0, 4, 8, 12]
    for k, r in enumerate(reps[1:], 1):
        assert r.frame == k and r.vo_valid == 1 and r.tracked_from_last_frame >= 60
        assert abs(np.array(r.vo_pose[:]).reshape(3, 4)[0, 3] - 0.54) < 0.05            # the camera moves 0.54 m to its right per frame
    r = reps[8]
    assert r.checked_for_kf and r.vo_force_new_kf and r.n_query >= 1 and r.query_ids[0] in (0, 1) and r.similar_kfs[0] == 1
    assert r.kf_id == 2 and r.n_created_edges == 1 and r.total_sqr_error_final <= r.total_sqr_error_init and r.n_keyframes == 3
    assert slam.counts()[0] == 4 and slam.counts()[1] == 3 and slam.counts()[2] > 300
    out = os.path.join(tmp_path, "out_kf_poses.txt")
    X, info = slam.optimize_pose_graph(out)
    assert X.shape == (4, 12) and info.stop_reason == 1                                  # a tree: nothing to move
    assert abs(X[3].reshape(3, 4)[1, 3] + 12 * 0.54) < 0.1                               # 12 baselines to the right = robot -y
    assert np.loadtxt(out).shape == (4, 7)
    with pytest.raises(capi.SskError, match="fixed once"):
        slam._chk(slam.lib.ssk_slam_set_option(slam.h_, b"submap_size", 5.0))
    slam.close()


def test_slam_abi_needs_a_vocabulary(pkg, ctx):
    from ssk_amd import capi
    slam = capi.Slam(ctx, synth.W, synth.H)
    l, r = synth.synth_stereo_pair(1)
    with pytest.raises(capi.SskError, match="no vocabulary"):
        slam.process_frame(l, r)
    slam.close()
