#!/usr/bin/env python3
"""Diagnostic: the forward drives of tests/test_forward_motion_gpu.py over several seeds: keyframes, their tracked counts, recovered path."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth, bow_vocab as bv
ctx = pkg.Context(0)
prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
for seed in [int(a) for a in sys.argv[1:]] or [1]:
    for yaw in (0.0, np.deg2rad(0.8)):
        frames, poses = synth.synth_forward_sequence(seed, 13, step=0.5, yaw_per_frame=yaw)
        tmp = tempfile.mkdtemp()
        voc = os.path.join(tmp, "voc.yml.gz")
        bv.write_yaml(bv.train([ctx.frontend_run(l, r, prm)["desc_left"] for l, r in frames[::3]], k=8, L=3, seed=5), voc)
        slam = capi.Slam(ctx, synth.W, synth.H, "kitti")
        slam.load_vocabulary(voc)
        kfs = []
        for k, (l, r) in enumerate(frames):
            rp = slam.process_frame(l, r)
            if rp.inserted_kf:
                kfs.append((k, rp.highest_tracked, rp.tracked_from_last_kf))
        X, info = slam.optimize_pose_graph()
        P = X.reshape(-1, 3, 4)[:, :, 3]
        S = np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
        err = [float(np.linalg.norm(P[i] - S @ poses[k][1])) for i, (k, _, _) in enumerate(kfs)]
        print("seed", seed, "yaw %.3f" % yaw, "keyframes (frame, DA tracked, VO tracked)", kfs, "position error", np.round(err, 3))
        slam.close()
