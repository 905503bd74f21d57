#!/usr/bin/env python3
"""Host cost of ssk_lm_batch_create for the bench's LM batch (1024 windows, 64 distinct, E = 10) and for the depth-5 batch: wall time of the call,
the symbolic part and its threads (SSK_LM_BUILD_THREADS).  usage: python tools/lm_create_time.py [n_windows] [depth]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = pkg.Context(0)
nd = 64 if depth == 2 else 4
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=depth, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(nd)]
batch = [wins[k % nd] for k in range(n)]
mar = capi.LmBatch.marshal(batch)       # (the C view of the windows, once)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for rep in range(reps):
    t0 = time.perf_counter()
    b = capi.LmBatch(ctx, batch, marshalled=mar)
    dt = 1e3 * (time.perf_counter() - t0)
    ms, thr = b.symbolic_ms()
    print("create %d windows (E %d): %.1f ms wall (windows marshalled once, outside), symbolic %.1f ms on %d threads = %.4f ms per window" % (n, wins[0].E, dt, ms, thr, ms / n))
    b.close()
ctx.close()
