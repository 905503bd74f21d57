"""Host-side keyframe policy (host/ssk_kf_policy.h) and vocabulary file reader (host/ssk_bow_vocab_io.h) against the
oracle / the test-side parser, through the CPU-only policy_tool driver.  No GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import bow_vocab as bv
import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ssk_build", os.path.join(ROOT, "srba-stereo-slam_amd", "build.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.build_host(verbose=False)
    return os.path.join(ROOT, "srba-stereo-slam_amd", "host", "policy_tool")


def run_policy(tool, lines):
    r = subprocess.run([tool, "policy"], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    return [l.split() for l in r.stdout.strip().split("\n")]


def _i64(a):
    return np.ascontiguousarray(a, np.int64)


def test_threshold_updates_match_oracle(tool):
    lib = oracle_lib.load().lib
    lib.ssko_update_translation_threshold.restype = C.c_double
    lib.ssko_update_rotation_threshold.restype = C.c_double
    lib.ssko_update_translation_threshold.argtypes = [C.c_double, C.c_double]
    lib.ssko_update_rotation_threshold.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(1)
    cases = [(float(x), float(th)) for x, th in zip(rng.uniform(-5, 120, 200), rng.integers(1, 120, 200))] + [(0.0, 25.0), (25.0, 25.0), (85.0, 85.0)]
    out = run_policy(tool, ["thr %r %r" % c for c in cases])
    for (x, th), o in zip(cases, out):
        assert float(o[0]) == lib.ssko_update_translation_threshold(x, th) and float(o[1]) == lib.ssko_update_rotation_threshold(x, th)
    # known answers: UTIL.cpp:153-171 clamps to [0.02, 0.3] m and [2, 15] deg
    assert lib.ssko_update_translation_threshold(0, 25) == 0.02 and lib.ssko_update_translation_threshold(25, 25) == 0.02 + 0.25
    assert lib.ssko_update_translation_threshold(100, 25) == 0.3 and lib.ssko_update_rotation_threshold(85, 85) == 15
    assert lib.ssko_update_rotation_threshold(0, 85) == 2


def test_vo_check_matches_oracle(tool):
    lib = oracle_lib.load().lib
    lib.ssko_vo_check.argtypes = [C.c_double] * 4 + [C.c_int64] + [C.c_double] * 4 + [C.c_int, C.c_void_p]
    rng = np.random.default_rng(2)
    lines, exp = [], []
    for _ in range(300):
        a = [float(v) for v in rng.uniform(0, 3, 2)] + [float(v) for v in rng.uniform(0, 0.6, 2)]
        a = [a[0], a[2], a[1], a[3]]
        tr, th = int(rng.integers(0, 120)), int(rng.choice([0, 40]))
        lim = [2.0, 25.0, float(rng.choice([0.02, 0.3, 4.0])), float(rng.choice([2.0, 15.0, 50.0]))]
        lines.append("vocheck %r %r %r %r %d %r %r %r %r %d" % (a[0], a[1], a[2], a[3], tr, lim[0], lim[1], lim[2], lim[3], th))
        out = np.zeros(4, np.int32)
        lib.ssko_vo_check(a[0], a[1], a[2], a[3], tr, lim[0], lim[1], lim[2], lim[3], th, out.ctypes.data)
        exp.append(out.tolist())
    got = [[int(x) for x in o] for o in run_policy(tool, lines)]
    assert got == exp and any(e[0] for e in exp) and any(e[1] for e in exp) and any(not e[3] for e in exp)


def oracle_similar(lib, new_id, q, submap, lcd):
    ids, sc, topo = _i64([x[0] for x in q]), np.ascontiguousarray([x[1] for x in q], np.float64), _i64([x[2] for x in q])
    sim = np.zeros(len(q) + 1, np.int64)
    n, lc = C.c_int(), C.c_int64()
    lib.ssko_get_similar_kfs.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    pot = lib.ssko_get_similar_kfs(new_id, len(q), ids.ctypes.data, sc.ctypes.data, topo.ctypes.data, submap, lcd, sim.ctypes.data, C.byref(n), C.byref(lc))
    return pot, lc.value, sim[:n.value].tolist()


def test_similar_kfs_matches_oracle_and_known_answers(tool):
    lib = oracle_lib.load().lib
    rng = np.random.default_rng(3)
    lines, exp = [], []
    for _ in range(300):
        new_id = int(rng.integers(2, 400))
        nq = int(rng.integers(1, 5))
        ids = rng.choice(new_id, nq, replace=False)
        if rng.random() < 0.5:
            ids[rng.integers(nq)] = new_id - 1
            ids = np.array(list(dict.fromkeys(ids.tolist())))
        sc = np.sort(rng.uniform(0.0, 0.6, len(ids)))[::-1]
        q = [(int(i), float(s), int(rng.choice([-1, 0, 1, 3, 9, 11, 40]))) for i, s in zip(ids, sc)]
        lines.append("similar %d 10 10 %d " % (new_id, len(q)) + " ".join("%d %r %d" % x for x in q))
        exp.append(oracle_similar(lib, new_id, q, 10, 10))
    out = run_policy(tool, lines)
    for o, e in zip(out, exp):
        assert (int(o[0]), int(o[1]), [int(x) for x in o[3:]]) == e and int(o[2]) == len(e[2])
    assert any(e[0] for e in exp) and any(len(e[2]) > 2 for e in exp)
    # keyframe 57: previous always first; 0.31 > 0.8*0.35 joins; keyframe 3 is far (not in the tree) and scores > 0.05: loop closure
    assert oracle_similar(lib, 57, [(56, 0.35, 1), (55, 0.31, 2), (3, 0.09, -1), (4, 0.08, -1)], 10, 10) == (1, 3, [56, 55, 3])
    assert oracle_similar(lib, 57, [(56, 0.35, 1)], 10, 10) == (0, -1, [56])                        # a single result: nothing to analyse
    assert oracle_similar(lib, 57, [(50, 0.35, 0), (3, 0.04, -1)], 10, 10) == (0, -1, [56, 50])      # far but below 0.05
    assert oracle_similar(lib, 57, [(40, 0.35, 9), (41, 0.2, 10), (42, 0.2, 11)], 10, 10) == (1, 42, [56, 40, 42])   # distance must EXCEED lc_distance


def oracle_decide(lib, ins, da, pot, lc, vo, uth, upl, mt, mr):
    kf, tr = _i64([d[0] for d in da]), _i64([d[1] for d in da])
    th = np.array([mt, mr])
    srt = np.zeros(max(len(da), 1), np.int64)
    hi = C.c_int64()
    fl = np.zeros(3, np.int32)
    lib.ssko_kf_decision.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    lib.ssko_kf_decision(ins, len(da), kf.ctypes.data, tr.ctypes.data, pot, lc, vo, uth, upl, th.ctypes.data, srt.ctypes.data, C.byref(hi), fl.ctypes.data)
    return fl.tolist(), hi.value, th.tolist(), srt[:len(da)].tolist()


def test_insertion_decision_matches_oracle_and_known_answers(tool):
    lib = oracle_lib.load().lib
    rng = np.random.default_rng(4)
    lines, exp = [], []
    for _ in range(400):
        n = int(rng.integers(1, 6))
        da = [(int(k), int(t)) for k, t in zip(rng.choice(300, n, replace=False), rng.choice([0, 10, 59, 60, 61, 85, 86, 200, 200, 30], n))]
        pot = int(rng.random() < 0.4)
        lc = int(da[rng.integers(n)][0]) if pot and rng.random() < 0.8 else (int(rng.integers(300, 400)) if pot else -1)
        ins, vo = int(rng.random() < 0.2), int(rng.random() < 0.2)
        lines.append("decide %d %d %d %d 60 25 4.0 50.0 %d " % (ins, pot, lc, vo, n) + " ".join("%d %d" % d for d in da))
        exp.append(oracle_decide(lib, ins, da, pot, lc, vo, 60, 25, 4.0, 50.0))
    out = run_policy(tool, lines)
    for o, e in zip(out, exp):
        assert [int(o[0]), int(o[1]), int(o[2])] == e[0] and int(o[3]) == e[1] and [float(o[4]), float(o[5])] == e[2]
        assert [int(x) for x in o[7:]] == e[3]
    assert any(e[0][1] for e in exp) and any(e[0][2] for e in exp) and any(not e[0][0] for e in exp)
    # 61 tracked features with updated_matches_th 60 (+25): no keyframe, thresholds shrink to 0.02 + 0.25/25*1 m and 15 + 13/85*(61-85) deg
    f, hi, th, srt = oracle_decide(lib, 0, [(7, 61), (3, 20)], 0, -1, 0, 60, 25, 4.0, 50.0)
    assert f == [0, 0, 1] and hi == 61 and th == [0.02 + (0.25 / 25) * 1, 15 + 13 / 85 * (61 - 85)] and srt == [0, 1]
    assert oracle_decide(lib, 0, [(7, 59)], 0, -1, 0, 60, 25, 4.0, 50.0)[0] == [1, 0, 0]           # below the threshold: insert
    assert oracle_decide(lib, 0, [(7, 200)], 0, -1, 0, 60, 25, 4.0, 50.0)[0] == [0, 0, 0]          # plenty tracked: nothing changes
    # loop closure: keyframe 3 has more than half of the best count -> confirmed, inserted, and moved to the front
    f, hi, th, srt = oracle_decide(lib, 0, [(56, 200), (55, 150), (3, 101)], 1, 3, 0, 60, 25, 4.0, 50.0)
    assert f == [1, 1, 0] and srt == [2, 0, 1]
    assert oracle_decide(lib, 0, [(56, 200), (3, 100)], 1, 3, 0, 60, 25, 4.0, 50.0)[0] == [0, 0, 0]   # exactly half: not confirmed, and no
    # threshold adaptation on this branch either (EST.cpp:481-521)


def _read_packed(path):
    raw = open(path, "rb").read()
    k, L, sc, we, n, nw = np.frombuffer(raw, np.int32, 6)
    o = 24
    nid = np.frombuffer(raw, np.int32, n, o); o += 4 * n
    pid = np.frombuffer(raw, np.int32, n, o); o += 4 * n
    w = np.frombuffer(raw, np.float64, n, o); o += 8 * n
    d = np.frombuffer(raw, np.uint8, 32 * n, o).reshape(n, 32); o += 32 * n
    wn = np.frombuffer(raw, np.int32, nw, o)
    return dict(k=int(k), L=int(L), scoring=int(sc), weighting=int(we), node_id=nid, parent_id=pid, weight=w, desc=d, word_node=wn)


@pytest.mark.parametrize("gz", [True, False])
def test_vocabulary_reader_on_synthetic_file(tool, tmp_path, gz):
    v = bv.synth(k=6, L=3, seed=17, ragged=True)
    path = str(tmp_path / ("voc.yml.gz" if gz else "voc.yml"))
    bv.write_yaml(v, path)
    out = str(tmp_path / "voc.bin")
    subprocess.run([tool, "vocab", path, out], check=True)
    g = _read_packed(out)
    for key in v:
        assert np.array_equal(np.asarray(v[key]), np.asarray(g[key])), key


def test_vocabulary_reader_refuses_bad_files(tool, tmp_path):
    p = str(tmp_path / "x.yml")
    open(p, "w").write("%YAML:1.0\nfoo: 1\n")
    r = subprocess.run([tool, "vocab", p, str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "not a DBoW2 vocabulary" in r.stderr
    v = bv.synth(k=2, L=1, seed=1)
    bv.write_yaml(v, p)
    import re
    s = re.sub(r'descriptor:"[01]', 'descriptor:"', open(p).read(), count=1)      # a 255-bit descriptor
    open(p, "w").write(s)
    r = subprocess.run([tool, "vocab", p, str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 3 and "256 bits" in r.stderr
    r = subprocess.run([tool, "vocab", str(tmp_path / "missing.yml.gz"), str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 3


@pytest.mark.skipif(not bv.have_reference_vocabulary(), reason="the reference's vocabulary file is only in the build container")
def test_vocabulary_reader_on_the_reference_file(tool, tmp_path):
    out = str(tmp_path / "voc.bin")
    subprocess.run([tool, "vocab", bv.REF_VOC, out], check=True)
    g, v = _read_packed(out), bv.parse(bv.REF_VOC)
    assert (g["k"], g["L"], g["scoring"], g["weighting"]) == (8, 5, 0, 0)
    for key in v:
        assert np.array_equal(np.asarray(v[key]), np.asarray(g[key])), key
