"""Timing helper (not a test): the global pose-graph optimisation at KITTI-00 size -- the C-ABI on the GPU beside the CPU
oracle on the same graph.  usage: python tests/pg_timing.py [n_nodes] [n_closures]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge   # noqa: E402
import synth_pg                # noqa: E402
import oracle_lib              # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1136
    nc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    g = synth_pg.make_loop(5, n=n, radius=400.0, noise_t=0.02, noise_r=0.0005, n_closures=nc, edge_forward=False)
    pkg = ge.load_package()
    from ssk_amd import capi
    ctx = pkg.Context(0)
    prm = capi.PgParams()
    prm.max_iters = 100; prm.pcg_max_iters = 1000; prm.tau = 1e-3; prm.max_lambda = 1e20; prm.converged_rel_decrease = 1e-10; prm.pcg_tol = 1e-10
    for rep in range(2):
        X = g["init"].copy()
        t = time.perf_counter()
        Xg, rg = ctx.posegraph_optimize(X, g["edge_ij"], g["edge_z"], 0, prm)
        tg = time.perf_counter() - t
    print("gpu: %.3f s  iters %d accepted %d stop %d cg %d cost %.6g -> %.6g" % (tg, rg.iters, rg.accepted, rg.stop_reason, rg.pcg_iters_total, rg.cost_init, rg.cost_final))
    o = oracle_lib.load()
    op = oracle_lib.PgParams()
    for f in ("max_iters", "pcg_max_iters", "tau", "max_lambda", "converged_rel_decrease", "pcg_tol"):
        setattr(op, f, getattr(prm, f))
    t = time.perf_counter()
    Xo, ro = o.pg_optimize(g["init"].copy(), g["edge_ij"], g["edge_z"], 0, op)
    to = time.perf_counter() - t
    print("cpu oracle: %.3f s  iters %d accepted %d stop %d cg %d cost %.6g -> %.6g" % (to, ro.iters, ro.accepted, ro.stop_reason, ro.pcg_iters_total, ro.cost_init, ro.cost_final))
    print("max |X_gpu - X_cpu| = %.3g" % np.abs(np.asarray(Xg) - np.asarray(Xo)).max())


if __name__ == "__main__":
    main()
