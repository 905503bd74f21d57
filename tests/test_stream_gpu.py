"""GPU: the whole frame loop of the facade (CSRBAStereoSLAMEstimator::performStereoSLAM, EST.cpp:44-937) on a synthetic
sequence -- odometry between frames, keyframe check, bag-of-words query on the device, similar keyframes, loop-closure
detection and confirmation, keyframe insertion.  BASELINE.json configs[3]-shaped plumbing at test size.  The vocabulary is
trained here (hierarchical k-medians, as DBoW2 builds one) on descriptors the device front-end extracts from the sequence
and written in the DBoW2 file format (the reference's own file cannot travel to the GPU box)."""
import os
import subprocess

import numpy as np
import pytest

import bow_vocab as bv
import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "srba-stereo-slam_amd", "host", "facade_demo")


def run_stream(ctx, tmp_path, frames, opts, train_on):
    from ssk_amd import capi
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    voc = os.path.join(tmp_path, "voc.yml.gz")
    bv.write_yaml(bv.train([ctx.frontend_run(l, r, prm)["desc_left"] for l, r in train_on], k=8, L=3, seed=5), voc)
    for k, (l, r) in enumerate(frames):
        l.tofile(os.path.join(tmp_path, "left_%04d.raw" % k))
        r.tofile(os.path.join(tmp_path, "right_%04d.raw" % k))
    cmd = [DEMO, "stream", voc, str(tmp_path), str(len(frames)), str(synth.W), str(synth.H)] + ["%s=%s" % kv for kv in opts.items()]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr + p.stdout
    if os.environ.get("SSK_TEST_DUMP"):
        open(os.environ["SSK_TEST_DUMP"], "a").write(p.stdout)
    rows = []
    for line in p.stdout.splitlines():
        if not line.startswith("FRAME "):
            continue
        t = line.split()
        d = {t[i]: t[i + 1] for i in range(0, t.index("query"), 2)}
        q = t[t.index("query") + 1:t.index("similar")]
        d["query"] = [(int(x.split(":")[0]), float(x.split(":")[1])) for x in q]
        rest = t[t.index("similar") + 1:]
        d["similar"] = [int(x) for x in (rest[:rest.index("kf")] if "kf" in rest else rest)]
        rows.append(d)
    return rows, p.stdout


def test_every_frame_a_keyframe_and_loop_closure(pkg, ctx, tmp_path):
    """Limits below the per-frame motion (4 baselines = 2.16 m): odometry forces a keyframe at every frame.  The scene is near
    (<= 12 m), so nothing stays in view for more than 10 frames and the keyframe graph is a chain of local maps.  After 44 keyframes the camera is put back at places 2, 3, 4 of the start:
    no overlap with place 43 (odometry invalid, frame skipped), then the query finds the old keyframes, the best one joins the
    similar keyframes, the next one -- outside the current local map's spanning tree -- is the potential loop closure and the
    tracked features confirm it."""
    assert os.path.exists(DEMO), "host facade not built: python __graft_entry__.py build"
    n = 44
    frames, _ = synth.synth_translation_sequence(5, n, z_max=12.0, step=4)
    # the same places under other (stronger) sensor noise.  (sigma 2: at sigma 4 the scores of the keyframes around the revisited place
    # tie within 0.003 -- 0.5095 / 0.5067 -- and a one-bit change of a few descriptors decides whether the potential loop closure is the
    # neighbouring place, confirmed by its tracked features, or one two places away, which shares too little to be confirmed)
    again, _ = synth.synth_translation_sequence(5, n, z_max=12.0, step=4, noise=2.0)
    seq = frames + [again[2], again[3], again[4]]
    rows, out = run_stream(ctx, tmp_path, seq, dict(max_translation=0.3, submap_size=4, max_tree_depth=3, max_optimize_depth=3, lc_distance=3),
                           train_on=frames[::4])
    assert len(rows) == len(seq)
    for k in range(1, n):
        r = rows[k]
        assert r["vo_valid"] == "1" and int(r["tracked_frame"]) >= 60 and abs(float(r["vo_t"]) - 2.16) < 0.1, (k, r)
        assert r["force_kf"] == "1" and r["checked"] == "1" and r["inserted"] == "1" and int(r["n_kf"]) == k + 1
        ids = [q[0] for q in r["query"]]
        assert ids[0] >= k - 3 and k - 1 in ids and r["similar"][0] == k - 1, (k, r)   # the nearest keyframes score best; the previous one is always similar
        assert r["conf_lc"] == "0", (k, r)
    # the jump back: place 2 right after place 43 -- no overlap, odometry invalid, frame skipped (EST.cpp:318-323)
    assert rows[n]["vo_valid"] == "0" and rows[n]["checked"] == "0" and int(rows[n]["n_kf"]) == n
    # from place 2 to place 3: odometry works again; the query returns the old keyframes around place 3
    r = rows[n + 1]
    assert r["vo_valid"] == "1" and r["checked"] == "1"
    assert r["query"][0][0] == 3 and r["query"][0][1] > 0.6 and {q[0] for q in r["query"]} <= set(range(8)), r
    assert r["pot_lc"] == "1" and r["conf_lc"] == "1" and r["inserted"] == "1" and int(r["lc_id"]) in (0, 1, 2, 4, 5, 6), r
    assert r["similar"][:2] == [n - 1, 3] and int(r["lc_id"]) in r["similar"] and int(r["highest"]) >= 150
    assert "keyframes %d " % (n + 2) in out and "db %d" % (n + 2) in out
    # after the loop: the global pose graph (EST.cpp:945-987).  The loop-closure keyframe hangs on the old local map only, so
    # the graph is a tree (cost 0 at the composed poses) and the optimiser has nothing to move; out_kf_poses.txt is written in
    # the reference's format.  Keyframe k of the first pass sits 4 k baselines to the right of keyframe 0 (robot -y).
    pg = [l.split() for l in out.splitlines() if l.startswith("posegraph ")][0]
    assert int(pg[2]) == n + 2 and int(pg[8]) == 1 and float(pg[10]) < 1e-12 and float(pg[12]) < 1e-12, pg
    poses = np.loadtxt(os.path.join(tmp_path, "out_kf_poses.txt"))
    assert poses.shape == (n + 2, 7) and np.array_equal(poses[:, 0], np.arange(n + 2))
    for k in (1, 10, n - 1):
        assert abs(poses[k, 2] + 2.16 * k) < 0.03 * 2.16 * k + 0.05 and abs(poses[k, 1]) < 0.5 and abs(poses[k, 3]) < 0.5, (k, poses[k])
    # the revisit keyframes come back to places 3 and 4
    assert abs(poses[n, 2] + 2.16 * 3) < 0.5 and abs(poses[n + 1, 2] + 2.16 * 4) < 0.5, poses[n:]


def test_loop_closure_under_stronger_noise(pkg, ctx, tmp_path):
    """The revisit of test_every_frame_a_keyframe_and_loop_closure at sensor noise sigma 4 (the scenario of rounds 1-2): there the scores of the
    keyframes around the revisited place tie within 0.003, so WHICH old keyframe becomes the potential loop closure is decided by single descriptor
    bits.  What must hold whatever the tie's outcome: the query's best results are the old keyframes around place 3, a potential loop closure
    with one of them is raised, and when it is a direct neighbour of the revisited place (which shares most of its view) it is confirmed and
    the keyframe inserted."""
    assert os.path.exists(DEMO), "host facade not built: python __graft_entry__.py build"
    n = 44
    frames, _ = synth.synth_translation_sequence(5, n, z_max=12.0, step=4)
    again, _ = synth.synth_translation_sequence(5, n, z_max=12.0, step=4, noise=4.0)
    seq = frames + [again[2], again[3], again[4]]
    rows, out = run_stream(ctx, tmp_path, seq, dict(max_translation=0.3, submap_size=4, max_tree_depth=3, max_optimize_depth=3, lc_distance=3),
                           train_on=frames[::4])
    assert rows[n]["vo_valid"] == "0" and int(rows[n]["n_kf"]) == n
    r = rows[n + 1]
    assert r["vo_valid"] == "1" and r["checked"] == "1"
    assert r["query"][0][0] in (2, 3, 4) and {q[0] for q in r["query"]} <= set(range(8)), r
    assert r["pot_lc"] == "1" and int(r["lc_id"]) in (0, 1, 2, 4, 5, 6), r
    if abs(int(r["lc_id"]) - 3) <= 1:
        assert r["conf_lc"] == "1" and r["inserted"] == "1", r


def test_check_without_insertion(pkg, ctx, tmp_path):
    """KITTI limits (2 m / 25 deg): a keyframe is forced every 4th frame (4 x 0.54 m > 2 m); in between, the loop only
    accumulates odometry."""
    frames, _ = synth.synth_translation_sequence(6, 13)
    rows, out = run_stream(ctx, tmp_path, frames, dict(), train_on=frames[::3])
    inserted = [k for k, r in enumerate(rows) if r["inserted"] == "1"]
    assert inserted == [0, 4, 8, 12], inserted
    for k, r in enumerate(rows):
        assert r["vo_valid"] == "1"
        if k not in inserted:
            assert r["checked"] == "0" and r["force_kf"] == "0"
    assert "keyframes 4 " in out
