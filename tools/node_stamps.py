import ctypes as C, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(8)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 8] for k in range(n)])
for rep in range(2):
    b.reset(); ctx.synchronize()
    t0 = time.perf_counter(); b.run(lp); r = b.results(); dt = time.perf_counter() - t0
print("E %d obs %d: %d windows in %.3f ms, iters %s" % (wins[0].E, len(wins[0].obs_lm), n, 1e3 * dt, [x.iters for x in r[:8]]))
b.close(); ctx.close()
