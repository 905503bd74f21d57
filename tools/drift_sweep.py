#!/usr/bin/env python3
"""Seed sweep of the full-length stream (BASELINE configs[3], tests/test_config3_gpu.py): drift after the global pose graph for several frame-source
seeds and both ORB steering forms (SSK_ORB_STEERING=1 cv::ORB's, 0 exact centroid direction), as a table (profiles/r04_drift_sweep.json).
usage: python tools/drift_sweep.py [n_frames] [n_back] [seeds...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tools")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import stream_full
n, nb, seed = %(n)d, %(nb)d, %(seed)d
o = stream_full.run(n, nb, seed=seed, verbose=False)
X = o["poses"].reshape(-1, 3, 4)
truth = np.zeros((len(o["kf_pos"]), 3)); truth[:, 1] = -0.54 * np.array(o["kf_pos"])
path = 0.54 * (n - 1)
err = np.linalg.norm(X[:, :, 3] - truth, axis=1)
along = np.abs(X[:, 1, 3] - truth[:, 1])
print(json.dumps({"seed": seed, "along_pct": 100 * along.max() / path, "err3d_pct": 100 * err.max() / path, "heading": float(np.abs(X[:, :, :3] - np.eye(3)).max()),
                  "invalid": len(o["invalid_odometry"]), "multi_edge_kfs": o["multi_edge_kfs"], "frames_per_s": o["frames_per_s"], "ms_per_kf_frame": o["ms_per_keyframe_frame"]}))
'''
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 160
seeds = [int(x) for x in sys.argv[3:]] or [9, 10, 11, 12, 13]
out = {"frames": n, "n_back": nb, "runs": []}
for steer in (1, 0):
    for seed in seeds:
        r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT, n=n, nb=nb, seed=seed)], env=dict(os.environ, SSK_ORB_STEERING=str(steer)), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        rec = json.loads(line[-1]) if line else {"seed": seed, "error": r.stderr[-300:]}
        rec["steering"] = steer
        out["runs"].append(rec)
        print(rec, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "drift_sweep.json"), "w"), indent=1)
