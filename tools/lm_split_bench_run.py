#!/usr/bin/env python3
"""The bench's LM batch (1 024 windows of 10 edges, 8 distinct) forced onto the phase-split kernels (SSK_LM_KERNEL=split), three launches: for
rocprofv3 --kernel-trace --stats (how far the split path is from the node kernel on small windows, and where)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SSK_LM_KERNEL"] = "split"
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(8)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 8] for k in range(n)])
print(b.kernel_name())
for rep in range(3):
    b.reset(); ctx.synchronize()
    t0 = time.perf_counter(); b.run(lp); r = b.results(); print("%.3f ms" % (1e3 * (time.perf_counter() - t0)))
b.close(); ctx.close()
