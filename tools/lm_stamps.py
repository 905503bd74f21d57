#!/usr/bin/env python3
"""Diagnostic: per-phase cycle stamps of the LM kernels on the bench window (build with SSK_EXTRA_CFLAGS=-DSSK_LM_STAMPS).
usage: SSK_EXTRA_CFLAGS=-DSSK_LM_STAMPS python tools/lm_stamps.py [n_windows]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
pkg = ge.load_package()
from ssk_amd import capi  # noqa: E402
import synth_lm as sl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = int(os.environ.get("LM_DEPTH", "2"))
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=depth, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(8)]
lp = capi.LmParams()
C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 8] for k in range(n)])
for rep in range(2):
    b.reset(); ctx.synchronize()
    t0 = time.perf_counter()
    b.run(lp)
    r = b.results()
    dt = time.perf_counter() - t0
print("E %d obs %d: %d windows in %.3f ms, iters %s" % (wins[0].E, len(wins[0].obs_lm), n, 1e3 * dt, [x.iters for x in r[:8]]))
b.close(); ctx.close()
