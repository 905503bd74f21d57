"""CPU checks of the bag-of-words oracle (oracle/oracle_bow.c, contract W1-W5): known answers on hand-built
vocabularies, a brute-force numpy restatement, and the reference's own vocabulary file when it is present."""
import os

import numpy as np
import pytest

import bow_vocab as bv
import oracle_lib

GOLD = os.path.join(os.path.dirname(__file__), "golden", "bow_case0.npz")


def _popc(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def brute_transform(v, feats):
    """numpy / pure-Python restatement of W2 + W3 straight from the record arrays."""
    kids = {}
    for i, (n, p) in enumerate(zip(v["node_id"], v["parent_id"])):
        kids.setdefault(int(p), []).append(i)
    word_of = {int(n): w for w, n in enumerate(v["word_node"])}
    acc = {}
    for f in feats:
        node = 0
        rec = None
        while node in kids:
            best, best_d = None, None
            for i in kids[node]:
                d = _popc(f, v["desc"][i])
                if best is None or d < best_d:
                    best, best_d = i, d
            rec = best
            node = int(v["node_id"][best])
        if rec is None:
            continue
        w, wt = word_of.get(node, -1), float(v["weight"][rec])
        if w >= 0 and wt > 0:
            acc[w] = acc[w] + wt if w in acc else wt
    words = sorted(acc)
    vals = [acc[w] for w in words]
    norm = 0.0
    for x in vals:
        norm += abs(x)
    if norm > 0:
        vals = [x / norm for x in vals]
    return np.array(words, np.int32), np.array(vals)


def brute_query(db, words, vals, max_results, max_id=-1):
    res = []
    q = dict(zip(words.tolist(), vals.tolist()))
    for e, (w, v) in enumerate(db):
        if not (e < max_id or max_id == -1):
            continue
        s, hit = 0.0, False
        for wi, vi in zip(w.tolist(), v.tolist()):          # ascending word order
            if wi in q:
                val = abs(q[wi] - vi) - abs(q[wi]) - abs(vi)
                s = s + val if hit else val
                hit = True
        if hit:
            res.append((s, e))
    res.sort()
    res = res[:max_results] if max_results > 0 else res
    return np.array([e for _, e in res], np.int32), np.array([-s / 2.0 for s, _ in res])


def test_known_answers():
    o = oracle_lib.load()
    # two-level tree, k = 2: children 1, 2 of the root; 3, 4 under 1; node 2 is a leaf itself
    z = np.zeros(32, np.uint8)
    ones = np.full(32, 255, np.uint8)
    a = z.copy(); a[0] = 0x0F
    b = z.copy(); b[0] = 0xF0
    v = dict(k=2, L=2, scoring=0, weighting=0, node_id=np.array([1, 2, 3, 4], np.int32), parent_id=np.array([0, 0, 1, 1], np.int32),
             weight=np.array([0.0, 2.0, 1.0, 3.0]), desc=np.array([z, ones, a, b]), word_node=np.array([3, 4, 2], np.int32))
    V = o.bow_vocab(v)
    assert V.word(z) == (0, 1.0)                  # tie between nodes 3 and 4 (distance 4 each): the first child wins
    assert V.word(b) == (1, 3.0)
    assert V.word(ones) == (2, 2.0)
    half = z.copy(); half[:16] = 255              # 128 bits set: tie at the root, first child (node 1) wins, then tie again
    assert V.word(half)[0] == 0
    w, x = V.transform(np.array([z, z, b, ones]))
    assert w.tolist() == [0, 1, 2]
    assert np.array_equal(x, np.array([2.0, 3.0, 2.0]) / 7.0)
    D = o.bow_db(3)
    assert D.add(w, x) == 0 and D.add(np.array([2], np.int32), np.array([1.0])) == 1 and len(D) == 2
    ids, sc = D.query(w, x, 4)
    assert ids.tolist() == [0, 1] and abs(sc[0] - 1.0) < 1e-15
    assert sc[1] == -(abs(2 / 7 - 1.0) - 2 / 7 - 1.0) / 2
    assert D.query(w, x, 4, max_id=1)[0].tolist() == [0]
    assert D.query(np.array([0], np.int32), np.array([1.0]), 4)[0].tolist() == [0]      # entry 1 shares no word: not reported
    assert D.query(w, x, 1)[0].tolist() == [0]


def test_empty_inputs():
    o = oracle_lib.load()
    V = o.bow_vocab(bv.synth(k=3, L=2, seed=1))
    w, x = V.transform(np.zeros((0, 32), np.uint8))
    assert len(w) == 0
    D = o.bow_db(V.n_words)
    assert D.query(w, x, 4)[0].size == 0
    D.add(w, x)
    assert len(D) == 1 and D.query(np.array([0], np.int32), np.array([1.0]), 4)[0].size == 0


@pytest.mark.parametrize("k,L,seed,ragged", [(8, 3, 7, False), (10, 2, 3, False), (3, 5, 11, False), (8, 4, 13, True)])
def test_against_brute_force(k, L, seed, ragged):
    o = oracle_lib.load()
    v = bv.synth(k=k, L=L, seed=seed, ragged=ragged)
    V = o.bow_vocab(v)
    rng = np.random.default_rng(seed)
    db, D = [], o.bow_db(V.n_words)
    for e in range(12):
        f = bv.keyframe_descriptors(v, rng, n=120)
        w, x = V.transform(f)
        bw, bx = brute_transform(v, f)
        assert np.array_equal(w, bw) and np.array_equal(x, bx)
        assert abs(x.sum() - 1) < 1e-12 and np.all(np.diff(w) > 0)
        db.append((w, x)); D.add(w, x)
    for e in (0, 5, 11):
        for mr, mid in ((4, -1), (3, e), (0, -1)):
            ids, sc = D.query(*db[e], max_results=mr, max_id=mid)
            bi, bs = brute_query(db, *db[e], mr, mid)
            assert np.array_equal(ids, bi) and np.array_equal(sc, bs)
    assert D.query(*db[5], 1)[0][0] == 5 and abs(D.query(*db[5], 1)[1][0] - 1.0) < 1e-14


def test_golden_fixture():
    """tests/golden/bow_case0.npz (made by tests/golden/make_golden.py from this oracle): vocabulary, descriptors, vectors, results."""
    o = oracle_lib.load()
    g = np.load(GOLD)
    v = {k[2:]: g[k] for k in g.files if k.startswith("v_")}
    V = o.bow_vocab(v)
    D = o.bow_db(V.n_words)
    for e in range(int(g["n_kf"])):
        w, x = V.transform(g["desc"][g["desc_off"][e]:g["desc_off"][e + 1]])
        assert np.array_equal(w, g["bow_word"][g["bow_off"][e]:g["bow_off"][e + 1]])
        assert np.array_equal(x, g["bow_val"][g["bow_off"][e]:g["bow_off"][e + 1]])
        ids, sc = D.query(w, x, 4)
        n = int(g["res_n"][e])
        assert np.array_equal(ids, g["res_ids"][e, :n]) and np.array_equal(sc, g["res_scores"][e, :n])
        D.add(w, x)


@pytest.mark.skipif(not bv.have_reference_vocabulary(), reason="the reference's vocabulary file is only in the build container")
def test_reference_vocabulary_file():
    """demo/voc.yml.gz, the data file the reference ships: shape facts and self-consistency of the descent."""
    o = oracle_lib.load()
    v = bv.parse(bv.REF_VOC)
    assert (v["k"], v["L"], v["scoring"], v["weighting"]) == (8, 5, 0, 0)
    assert len(v["node_id"]) == 36939 and len(v["word_node"]) == 32273
    V = o.bow_vocab(v)
    rng = np.random.default_rng(5)
    rec_of = {int(n): i for i, n in enumerate(v["node_id"])}
    same = 0
    for w in rng.choice(len(v["word_node"]), 200, replace=False):
        i = rec_of[int(v["word_node"][w])]
        same += V.word(v["desc"][i]) == (int(w), float(v["weight"][i]))
    assert same >= 170          # a leaf's own centre descends to that leaf (94 % over the file; medians of binary clusters)
    f = bv.keyframe_descriptors(v, rng, n=500, noise=20)
    w, x = V.transform(f)
    bw, bx = brute_transform(v, f[:60])
    assert np.array_equal(V.transform(f[:60])[0], bw) and np.array_equal(V.transform(f[:60])[1], bx)
    assert 0 < len(w) <= 500 and abs(x.sum() - 1) < 1e-12


def test_reference_vocabulary_subtree_fixture():
    """tests/golden/bow_refvoc_subtree.npz: the subtrees of the first two children of the root of the reference's vocabulary
    file (real BRIEF centres and idf weights, 9 274 nodes / 8 105 words; data, made by make_golden.py) -- travels to the GPU box."""
    o = oracle_lib.load()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_refvoc_subtree.npz"))
    v = {k[2:]: g[k] for k in g.files if k.startswith("v_")}
    assert int(v["k"]) == 8 and int(v["L"]) == 5 and len(v["node_id"]) == 9274
    V = o.bow_vocab(v)
    w, x = V.transform(g["desc_in"])
    assert np.array_equal(w, g["exp_word"]) and np.array_equal(x, g["exp_val"])
    bw, bx = brute_transform(v, g["desc_in"][:80])
    assert all(np.array_equal(a, b) for a, b in zip(V.transform(g["desc_in"][:80]), (bw, bx)))
