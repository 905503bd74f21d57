"""debug: one window through the split kernels vs the oracle; prints where the final state differs"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SSK_LM_KERNEL"] = "split"
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl, oracle_lib
orc = oracle_lib.load()
ctx = pkg.Context(0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
win, gt = sl.make_submap_problem(seed, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
w1 = sl.Window(np.where(win.edge_unknown_slot == win.E - 1, 0, -1), win.edge_inv_pose, -np.ones(len(win.lm_pos), np.int32), gt["gt_lm"], win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
w0 = sl.Window(-np.ones(len(win.edge_unknown_slot), np.int32), win.edge_inv_pose, win.lm_unknown_slot, win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
s2 = win.lm_unknown_slot.copy(); s2[::3] = -1; s2[s2 >= 0] = np.arange((s2 >= 0).sum())
w2 = sl.Window(win.edge_unknown_slot, win.edge_inv_pose, s2, win.lm_pos, win.obs_lm, win.obs_px, win.obs_path_off, win.path_edge)
for name, w in (("full", win), ("stage1", w1), ("lm only", w0), ("third held", w2)):
    prm = sl.make_params(robust=True)
    exp = orc.lm_solve(w, prm)
    lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(prm), C.sizeof(lp))
    b = capi.LmBatch(ctx, [w], trace=True)
    b.run(lp)
    r = b.results()[0]
    ent, dp, df = b.trace(0)
    poses, lms = b.state(0)
    er = exp["res"]
    d = np.abs(lms - exp["lms"]).max(axis=1)
    bad = np.flatnonzero(d > 1e-7)
    print(name, "iters", r.iters, er.iters, "acc", r.accepted, er.accepted, "stop", r.stop_reason, er.stop_reason, "max lm diff", d.max(), "bad lms", bad[:10], "slots", np.asarray(w.lm_unknown_slot)[bad[:10]])
    print("   gpu accepted", [t.accepted for t in ent], "\n   orc accepted", [t.accepted for t in exp["trace"]])
    if len(bad):
        l = bad[0]
        print("   lm", l, "n obs", int((np.asarray(w.obs_lm) == l).sum()), "gpu", lms[l], "orc", exp["lms"][l], "init", np.asarray(w.lm_pos).reshape(-1, 3)[l])
    b.close()
ctx.close()
