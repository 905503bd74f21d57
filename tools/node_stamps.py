import ctypes as C, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
"""Diagnostic: per-phase cycle stamps of lm_solve_node_kernel (window 0) -- library built by `bash tools/lm_variant.sh nstamps -DSSK_LM_STAMPS`, run with
SSK_LIB=_ab/libssk_nstamps.so.  usage: python tools/node_stamps.py [windows = 1024] [stage1: the stage-1 window of a depth-5 local area (new edge only, landmarks held)]"""
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
stage1 = len(sys.argv) > 2 and sys.argv[2] == "stage1"
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(8)]
if stage1:
    def st1(w):
        newest = int(np.flatnonzero(w.edge_unknown_slot == w.E - 1)[0])
        keep = [o for o in range(len(w.obs_lm)) if w.obs_path_off[o + 1] > w.obs_path_off[o] and (w.path_edge[w.obs_path_off[o]] >> 1) == newest]
        off, pe = [0], []
        for o in keep:
            pe.extend(w.path_edge[w.obs_path_off[o]:w.obs_path_off[o + 1]]); off.append(len(pe))
        return sl.Window(np.where(w.edge_unknown_slot == w.E - 1, 0, -1), w.edge_inv_pose, -np.ones(len(w.lm_pos), np.int32), w.lm_pos, w.obs_lm[keep], w.obs_px[keep], off, pe)
    wins = [st1(sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=5, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0]) for s in range(8)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k % 8] for k in range(n)])
for rep in range(2):
    b.reset(); ctx.synchronize()
    t0 = time.perf_counter(); b.run(lp); r = b.results(); dt = time.perf_counter() - t0
print("E %d obs %d: %d windows in %.3f ms, iters %s" % (wins[0].E, len(wins[0].obs_lm), n, 1e3 * dt, [x.iters for x in r[:8]]))
b.close(); ctx.close()
