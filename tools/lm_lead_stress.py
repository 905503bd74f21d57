#!/usr/bin/env python3
"""Stress of K2's leading-column path (ldl_leading_mfma + the tile solver: local areas beyond 255 unknowns): 20 launches of 256 windows of 70 optimised
edges (the full-length stream's 90th-percentile local area) must give the same iteration counts and final errors bit for bit."""
import ctypes as C, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=8, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(4)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 4] for k in range(256)])
assert b.solver_name() == "ldl_solve_mfma_lead", b.solver_name()
ref = None
t0 = time.time()
for rep in range(20):
    b.reset(); b.run(lp); r = b.results()
    sig = tuple((x.iters, x.err_final) for x in r)
    if ref is None: ref = sig
    assert sig == ref, "run %d differs" % rep
print("20 launches of %d windows of %d optimised edges (%s): identical results, %.1f s" % (len(r), wins[0].E, b.solver_name(), time.time() - t0))
b.close(); ctx.close()
