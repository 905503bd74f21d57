#!/usr/bin/env python3
"""Stress of the phase-split LM path: 40 launches of a mixed batch (1 024 depth-5 local areas + 24 small windows) must give the same iteration counts and
final errors bit for bit (flags and spin-waits inside K2, one owner and a fixed order per sum)."""
import ctypes as C, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(4)]
small = [sl.make_submap_problem(10 + i, n_kf=8 + 3 * i, n_lm=150 + 60 * i, submap_size=4 + i, obs_per_kf=40 + 10 * i)[0] for i in range(6)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 4] for k in range(1024)] + small * 4)
ref = None
t0 = time.time()
for rep in range(40):
    b.reset(); b.run(lp); r = b.results()
    sig = tuple((x.iters, x.err_final) for x in r)
    if ref is None: ref = sig
    assert sig == ref, "run %d differs" % rep
print("40 launches of %d windows: identical results, %.1f s" % (len(r), time.time() - t0))
b.close(); ctx.close()
