#!/usr/bin/env python3
"""Diagnostic: per-phase clock stamps of the MFMA-tile solver inside lm_split_solve_kernel (workgroup 0) on depth-5 windows (library built by
`bash tools/split_variant.sh stamps -DSSK_LM_STAMPS`; run with SSK_LIB=_ab/libssk_stamps.so).  usage: python tools/k2_stamps.py [windows]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(4)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 4] for k in range(n)])
out = (C.c_ulonglong * 16)()
lib = ctx.lib
b.run(lp); b.results(); lib.ssk_lm_k2_stamps(out, 1)
b.reset(); b.run(lp); b.results(); lib.ssk_lm_k2_stamps(out, 1)
cnt = max(out[15], 1)
names = {0: "tile wave 0: load", 1: "tile wave 0: wait for L'^-1 of the panel", 6: "tile wave 0: B", 2: "tile wave 0: barrier among the tile waves", 3: "tile wave 0: C",
         4: "tile wave 0: loop exit barrier", 5: "back-substitution (either)",
         7: "factor wave: wait for the diagonal tile", 8: "factor wave: elimination + L'^-1"}
tot = sum(out[i] for i in (0, 1, 2, 3, 4, 6)) + out[5] / 2
print("solves of workgroup 0: %d (6E = %d), clocks per solve (s_memtime units): %.0f" % (cnt, 6 * wins[0].E, tot / cnt))
for i, nm in names.items():
    print("  %-42s %9.0f  %5.1f %%" % (nm, out[i] / cnt, 100.0 * out[i] / tot))
b.close(); ctx.close()
