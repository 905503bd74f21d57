#!/usr/bin/env python3
"""Diagnostic: per-phase clock stamps of lm_split_linearise_kernel on the depth-5 batch (library built by tools/split_variant.sh stamps -DSSK_LM_STAMPS;
run with SSK_LIB=_ab/libssk_stamps.so).  Prints the average clocks of a tile workgroup per phase (thread 0's clock; 3 workgroups share a CU)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(4)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 4] for k in range(n)])
out = (C.c_ulonglong * 16)()
lib = ctx.lib
b.run(lp); b.results(); lib.ssk_lm_split_stamps(out, 1)
b.reset(); b.run(lp); b.results(); lib.ssk_lm_split_stamps(out, 1)
cnt = max(out[15], 1)
names = ["tile map + ctl + headers", "(unused)", "group + lane records", "group phase", "item fetch + barrier", "gradient items", "pair rounds + stores"]
tot = sum(out[i] for i in range(7))
print("K1 tile workgroups: %d, clocks per workgroup: %.0f" % (cnt, tot / cnt))
for i, nm in enumerate(names):
    print("  %-28s %9.0f  %5.1f %%" % (nm, out[i] / cnt, 100.0 * out[i] / tot))
b.close(); ctx.close()
