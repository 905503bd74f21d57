"""Golden FNV-1a hashes of every table the LM symbolic builder (srba-stereo-slam_amd/csrc/ssk_lm_node_build.h) emits for a set of seeded windows.
Written with the builder of round 5 (commit 240ddcd) BEFORE the round-6 rewrite of its loops: tests/test_lm_node_form.py::test_builder_tables_unchanged
holds every later builder to the same tables, bit for bit.  Usage: python tests/golden/make_node_table_hashes.py <liblm_node_emu.so> > node_table_hashes.json"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import synth_lm as sl

BIG = dict(n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))
CASES = [("d2", s, dict(BIG, opt_depth=2)) for s in range(2)] + [("d5", 0, dict(BIG, opt_depth=5))] + \
        [("small", 10 + i, dict(n_kf=8 + 3 * i, n_lm=150 + 60 * i, submap_size=4 + i, obs_per_kf=40 + 10 * i)) for i in range(6)] + \
        [("d3", i, dict(n_kf=60, n_lm=1500, submap_size=6, max_tree_depth=3, opt_depth=3, obs_per_kf=200)) for i in range(4)]


def hashes(lib, case):
    name, seed, kw = case
    w = sl.make_submap_problem(seed, **kw)[0]
    wc = w.c()
    out = {}
    for split in (0, 1):
        h = (C.c_ulonglong * 16)()
        rc = lib.nd_emu_tables_hash(C.byref(wc), 4, split, h)
        out[str(split)] = [int(rc)] + [int(v) for v in h]
    return out


if __name__ == "__main__":
    lib = C.CDLL(sys.argv[1])
    print(json.dumps({"%s/%d" % (c[0], c[1]): hashes(lib, c) for c in CASES}, indent=0))
