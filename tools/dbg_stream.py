#!/usr/bin/env python3
"""Diagnostic: the revisit frames of tests/test_stream_gpu.py under variations of the scenario (noise of the second pass, vocabulary seed)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
import synth, bow_vocab as bv
import test_stream_gpu as ts
ctx = pkg.Context(0)
n = 44
for noise in (2.0, 3.0, 4.0):
    for vseed in (5, 6):
        frames, _ = synth.synth_translation_sequence(5, n, z_max=12.0, step=4)
        again, _ = synth.synth_translation_sequence(5, n, z_max=12.0, step=4, noise=noise)
        seq = frames + [again[2], again[3], again[4]]
        tmp = tempfile.mkdtemp()
        orig = bv.train
        bv.train = lambda d, k, L, seed: orig(d, k=k, L=L, seed=vseed)
        rows, out = ts.run_stream(ctx, tmp, seq, dict(max_translation=0.3, submap_size=4, max_tree_depth=3, max_optimize_depth=3, lc_distance=3), train_on=frames[::4])
        bv.train = orig
        r = rows[n + 1]
        print("noise", noise, "vocab seed", vseed, "query", r["query"], "pot", r["pot_lc"], "conf", r["conf_lc"], "lc", r["lc_id"], "similar", r["similar"], "highest", r["highest"])
