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


# the STRUCTURE of four windows the estimator's stream builds (tests/test_stream_gpu.py::test_every_frame_a_keyframe_and_loop_closure, dumped by
# SSK_LM_DUMP on the GPU box: create calls 17, 47, 88, 89 -- local areas with thousands of one-observation landmarks, empty paths, a revisit): slots,
# observation -> landmark, paths.  Poses, positions and pixels do not shape the tables (pixels are copied: a ramp stands in).  The first round-6
# builder passed every seeded window above and failed on these (its whole-path cache took [2, 0] for the empty path).
EST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "estimator_windows.npz")
CASES += [("est", k, None) for k in (17, 47, 88, 89)]


def estimator_window(k):
    import numpy as np
    z = np.load(EST)
    g = lambda n: z["w%d_%s" % (k, n)]
    ne, nl, no = len(g("edge_slot")), len(g("lm_slot")), len(g("obs_lm"))
    return sl.Window(g("edge_slot"), np.tile(np.eye(3, 4).reshape(-1), ne), g("lm_slot"), np.zeros((nl, 3)), g("obs_lm"),
                     np.arange(4 * no, dtype=np.float32), g("path_off"), g("path_edge"))


def hashes(lib, case, threads=0):
    """threads > 0: the window's ranges of landmarks / tiles built by that many threads side by side (nd_team, round 6)."""
    name, seed, kw = case
    w = estimator_window(seed) if kw is None else sl.make_submap_problem(seed, **kw)[0]
    wc = w.c()
    out = {}
    for split in (0, 1):
        h = (C.c_ulonglong * 16)()
        rc = lib.nd_emu_tables_hash_team(C.byref(wc), 4, split, threads, h) if threads else lib.nd_emu_tables_hash(C.byref(wc), 4, split, h)
        out[str(split)] = [int(rc)] + [int(v) for v in h]
    return out


if __name__ == "__main__":
    lib = C.CDLL(sys.argv[1])
    print(json.dumps({"%s/%d" % (c[0], c[1]): hashes(lib, c) for c in CASES}, indent=0))
