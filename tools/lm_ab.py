#!/usr/bin/env python3
"""Same-box A/B of LM kernel builds: every library given (SSK_LIB=path, built by tools/lm_variant.sh) solves the bench batch in its own
process, the variants taken in turn `rounds` times; prints the best and the median time per launch of each.
usage: python tools/lm_ab.py [--windows 1024] [--rounds 3] [--depth 2] [--distinct 8] lib_a.so[:KEY=VALUE] lib_b.so ..."""
import argparse
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, sys, time
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import __graft_entry__ as ge
pkg = ge.load_package()
from ssk_amd import capi
import synth_lm as sl
n, depth = %(n)d, %(depth)d
ctx = pkg.Context(0)
wins = [sl.make_submap_problem(s, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=depth, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))[0] for s in range(%(distinct)d)]
lp = capi.LmParams(); C.memmove(C.byref(lp), C.byref(sl.make_params()), C.sizeof(lp))
b = capi.LmBatch(ctx, [wins[k %% %(distinct)d] for k in range(n)])
ts = []
for rep in range(6):
    b.reset(); ctx.synchronize()
    t0 = time.perf_counter(); b.run(lp); r = b.results(); ts.append(time.perf_counter() - t0)
print("%%.4f" %% (1e3 * min(ts[1:])))
b.close(); ctx.close()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    res = {l: [] for l in a.libs}
    code = CHILD % dict(root=ROOT, n=a.windows, depth=a.depth, distinct=a.distinct)
    for r in range(a.rounds):
        for l in a.libs:
            path, *kv = l.split(":")                   # lib.so[:KEY=VALUE ...]: the same library under different switches
            env = dict(os.environ, SSK_LIB=os.path.abspath(path), **dict(x.split("=", 1) for x in kv))
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            if out.returncode != 0:
                print(l, "FAILED", out.stderr[-400:]); continue
            res[l].append(float(out.stdout.strip().splitlines()[-1]))
    for l in a.libs:
        if res[l]:
            print("%-32s best %.3f ms  median %.3f ms  (%s)" % (os.path.basename(l), min(res[l]), statistics.median(res[l]), " ".join("%.3f" % x for x in res[l])))


if __name__ == "__main__":
    main()
