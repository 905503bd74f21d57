"""Diagnostic: the front-end run several times on one batch must give the same bytes every time (keypoints, descriptors, matches).
usage: python tools/fe_repeat.py [B] [runs]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ctx = pkg.Context(0)
b = capi.KfBatch(ctx, B, synth.W, synth.H, 512)
base = [synth.synth_stereo_pair(40 + i) for i in range(6)]
for i in range(B):
    b.upload(i, np.roll(base[i % 6][0], (i // 6, 5 * (i // 6)), axis=(0, 1)), np.roll(base[i % 6][1], (i // 6, 5 * (i // 6)), axis=(0, 1)))
fp = capi.FrontendParams(**synth.KITTI_FRONTEND)
ref = None
for r in range(runs):
    b.run_frontend(fp)
    got = [b.download(i) for i in range(B)]
    if ref is None: ref = got; continue
    bad = {"kp": 0, "desc": 0, "match": 0}
    for g, e in zip(got, ref):
        for k in ("kp_left", "kp_right"):
            bad["kp"] += int(len(g[k]) != len(e[k]) or any(not np.array_equal(g[k][f], e[k][f]) for f in g[k].dtype.names))
        for k in ("desc_left", "desc_right"):
            bad["desc"] += int(g[k].shape != e[k].shape or not np.array_equal(g[k], e[k]))
        bad["match"] += int(len(g["matches"]) != len(e["matches"]) or any(not np.array_equal(g["matches"][f], e["matches"][f]) for f in g["matches"].dtype.names))
    print("run", r, "differs from run 0 in", bad)
    if r == runs - 1:
        for i, (g, e) in enumerate(zip(got, ref)):
            for k in ("desc_left", "desc_right"):
                if g[k].shape == e[k].shape and not np.array_equal(g[k], e[k]):
                    rows = np.flatnonzero((g[k] != e[k]).any(axis=1))
                    x = np.bitwise_xor(g[k][rows], e[k][rows])
                    print("kf", i, k, "rows", rows[:8], "of", len(g[k]), "bits differing per row", [int(np.unpackbits(v).sum()) for v in x[:8]], "xy", [(int(g["kp_left" if k == "desc_left" else "kp_right"]["x"][q]), int(g["kp_left" if k == "desc_left" else "kp_right"]["y"][q])) for q in rows[:4]])
