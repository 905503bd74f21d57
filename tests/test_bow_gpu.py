"""GPU parity: bag-of-words place recognition (bow_words / bow_vector / bow_score / bow_topk kernels) vs the CPU oracle
through the C-ABI.  Word ids are integer work; the two f64 sums (L1 norm, L1 score) run in the oracle's order on the
device, so vectors, result ids and scores are compared bit for bit."""
import os

import numpy as np
import pytest

import bow_vocab as bv
import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "bow_case0.npz")


def both(pkg, ctx, oracle, v):
    from ssk_amd import capi
    return capi.BowVocab(ctx, v), oracle.bow_vocab(v)


@pytest.mark.parametrize("k,L,seed,n,ragged", [(8, 3, 7, 500, False), (10, 2, 3, 64, False), (3, 5, 11, 333, False), (12, 2, 4, 2048, False),
                                                (2, 7, 9, 1, False), (8, 5, 13, 500, True), (20, 3, 14, 300, True)])
def test_transform_matches_oracle(pkg, ctx, oracle, k, L, seed, n, ragged):
    v = bv.synth(k=k, L=L, seed=seed, ragged=ragged)
    G, O = both(pkg, ctx, oracle, v)
    assert G.info() == (k, L, len(v["node_id"]) + 1, len(v["word_node"]))
    rng = np.random.default_rng(seed)
    for f in (bv.keyframe_descriptors(v, rng, n=n), rng.integers(0, 256, (n, 32), dtype=np.uint8)):
        gw, gx = G.transform(f)
        ow, ox = O.transform(f)
        assert np.array_equal(gw, ow) and np.array_equal(gx, ox)


def test_transform_ties_and_stop_words(pkg, ctx, oracle):
    """All-equal children (every level a tie: the first child wins), repeated features (weights add one by one),
    zero-weight words (skipped), empty input."""
    v = bv.synth(k=4, L=3, seed=5, flip=0)                    # every node has the root's descriptor: ties everywhere
    G, O = both(pkg, ctx, oracle, v)
    f = np.random.default_rng(1).integers(0, 256, (40, 32), dtype=np.uint8)
    assert np.array_equal(G.transform(f)[0], O.transform(f)[0]) and len(G.transform(f)[0]) <= 1
    v = bv.synth(k=8, L=2, seed=6)
    G, O = both(pkg, ctx, oracle, v)
    leaf = np.flatnonzero(np.isin(v["node_id"], v["word_node"]))
    f = np.repeat(v["desc"][leaf[:9]], [1, 2, 3, 4, 5, 6, 7, 30, 100], axis=0)   # up to 100 occurrences of one word
    gw, gx = G.transform(f)
    ow, ox = O.transform(f)
    assert np.array_equal(gw, ow) and np.array_equal(gx, ox)
    zero = leaf[v["weight"][leaf] == 0]
    assert len(zero) > 0
    gw, gx = G.transform(v["desc"][zero])
    assert len(gw) == 0 and len(O.transform(v["desc"][zero])[0]) == 0
    assert len(G.transform(np.zeros((0, 32), np.uint8))[0]) == 0


def test_database_query_matches_oracle(pkg, ctx, oracle):
    from ssk_amd import capi
    v = bv.synth(k=8, L=3, seed=31)
    G, O = both(pkg, ctx, oracle, v)
    rng = np.random.default_rng(32)
    seq = bv.keyframe_sequence(v, rng, n_kf=40, n=200)
    seq += [seq[3].copy(), seq[17].copy()]
    gdb, odb = capi.BowDb(ctx, G, 64, 256), oracle.bow_db(O.n_words)
    for e, f in enumerate(seq):
        w, x = G.transform(f)
        for mr, mid in ((4, -1), (1, -1), (16, -1), (4, max(e - 5, 0))):
            gi, gs = gdb.query(w, x, mr, mid)
            oi, os_ = odb.query(w, x, mr, mid)
            assert np.array_equal(gi, oi) and np.array_equal(gs, os_), (e, mr, mid)
        assert gdb.add(w, x) == e and odb.add(w, x) == e and len(gdb) == e + 1
    gi, gs = gdb.query(*G.transform(seq[-1]), 4, 40)
    assert gi[0] == 17 and abs(gs[0] - 1) < 1e-14              # the revisit finds the old keyframe
    # a query without any common word reports nothing; so does an empty query
    far = np.setdiff1d(np.arange(G.n_words), np.concatenate([G.transform(f)[0] for f in seq]))[:3].astype(np.int32)
    assert len(gdb.query(far, np.full(len(far), 1.0 / max(len(far), 1)), 4)[0]) == 0
    assert len(gdb.query(np.zeros(0, np.int32), np.zeros(0), 4)[0]) == 0
    gdb.clear()
    assert len(gdb) == 0 and len(gdb.query(*G.transform(seq[0]), 4)[0]) == 0


def test_equal_scores_break_by_entry_id(pkg, ctx, oracle):
    from ssk_amd import capi
    v = bv.synth(k=8, L=2, seed=8)
    G, O = both(pkg, ctx, oracle, v)
    w, x = G.transform(bv.keyframe_descriptors(v, np.random.default_rng(2), n=50))
    gdb, odb = capi.BowDb(ctx, G, 8, 64), oracle.bow_db(O.n_words)
    for _ in range(6):
        gdb.add(w, x); odb.add(w, x)
    gi, gs = gdb.query(w, x, 4)
    oi, os_ = odb.query(w, x, 4)
    assert gi.tolist() == [0, 1, 2, 3] and np.array_equal(gi, oi) and np.array_equal(gs, os_)


def test_golden_fixture(pkg, ctx):
    from ssk_amd import capi
    g = np.load(GOLD)
    v = {k[2:]: g[k] for k in g.files if k.startswith("v_")}
    G = capi.BowVocab(ctx, v)
    db = capi.BowDb(ctx, G, 32, 256)
    for e in range(int(g["n_kf"])):
        w, x = G.transform(g["desc"][g["desc_off"][e]:g["desc_off"][e + 1]])
        assert np.array_equal(w, g["bow_word"][g["bow_off"][e]:g["bow_off"][e + 1]])
        assert np.array_equal(x, g["bow_val"][g["bow_off"][e]:g["bow_off"][e + 1]])
        ids, sc = db.query(w, x, 4)
        n = int(g["res_n"][e])
        assert np.array_equal(ids, g["res_ids"][e, :n]) and np.array_equal(sc, g["res_scores"][e, :n])
        db.add(w, x)


def test_batched_pipeline_matches_host_calls(pkg, ctx, oracle):
    """Device-resident path: front-end -> transform -> query -> insert on a batch, against the oracle fed with the
    descriptors the front-end produced."""
    from ssk_amd import capi
    v = bv.synth(k=8, L=4, seed=41)
    G, O = both(pkg, ctx, oracle, v)
    frames, _ = synth.synth_translation_sequence(77, 6)
    B = len(frames)
    prm = capi.FrontendParams(**synth.KITTI_FRONTEND)
    batch = capi.KfBatch(ctx, B, frames[0][0].shape[1], frames[0][0].shape[0], 500)
    for i, (l, r) in enumerate(frames):
        batch.upload(i, l, r)
    batch.run_frontend(prm)
    batch.bow_transform(G)
    gdb, odb = capi.BowDb(ctx, G, 16, 500), oracle.bow_db(O.n_words)
    vecs = []
    for i in range(B):
        kf = batch.download(i)
        ow, ox = O.transform(kf["desc_left"])
        gw, gx = batch.bow_download(i)
        assert len(ow) > 50 and np.array_equal(gw, ow) and np.array_equal(gx, ox)
        vecs.append((ow, ox))
        assert gdb.add_from_batch(batch, i) == i
        odb.add(ow, ox)
    # every keyframe against the entries before it (max_id = its own index), all in one launch
    batch.bow_query(gdb, 4, max_ids=np.arange(B))
    res = batch.bow_results()
    for i in range(B):
        oi, os_ = odb.query(*vecs[i], 4, i)
        assert np.array_equal(res[i][0], oi) and np.array_equal(res[i][1], os_), i
    assert len(res[0][0]) == 0 and len(res[B - 1][0]) == 4
    batch.bow_query(gdb, 2, kf_begin=2, kf_end=5)
    res = batch.bow_results(2, 5)
    for j, i in enumerate(range(2, 5)):
        oi, os_ = odb.query(*vecs[i], 2)
        assert np.array_equal(res[j][0], oi) and np.array_equal(res[j][1], os_)
    batch.close()


def test_errors_are_loud(pkg, ctx):
    from ssk_amd import capi
    v = bv.synth(k=4, L=2, seed=1)
    with pytest.raises(RuntimeError, match="TF_IDF"):
        capi.BowVocab(ctx, dict(v, weighting=1))
    with pytest.raises(RuntimeError, match="L1_NORM"):
        capi.BowVocab(ctx, dict(v, scoring=2))
    bad = dict(v, parent_id=v["parent_id"].copy())
    bad["parent_id"][3] = 10 ** 6
    with pytest.raises(RuntimeError, match="outside the tree"):
        capi.BowVocab(ctx, bad)
    G = capi.BowVocab(ctx, v)
    db = capi.BowDb(ctx, G, 2, 8)
    with pytest.raises(RuntimeError, match="ascending"):
        db.add(np.array([3, 1], np.int32), np.array([0.5, 0.5]))
    with pytest.raises(RuntimeError, match="stores 8"):
        db.add(np.arange(9, dtype=np.int32), np.full(9, 1 / 9))
    db.add(np.array([1], np.int32), np.array([1.0])); db.add(np.array([2], np.int32), np.array([1.0]))
    with pytest.raises(RuntimeError, match="full"):
        db.add(np.array([3], np.int32), np.array([1.0]))
    with pytest.raises(RuntimeError, match="max_results"):
        db.query(np.array([1], np.int32), np.array([1.0]), 0)


@pytest.mark.skipif(not bv.have_reference_vocabulary(), reason="the reference's vocabulary file is only in the build container")
def test_reference_vocabulary(pkg, ctx, oracle):
    v = bv.parse(bv.REF_VOC)
    G, O = both(pkg, ctx, oracle, v)
    f = bv.keyframe_descriptors(v, np.random.default_rng(3), n=500, noise=20)
    assert all(np.array_equal(a, b) for a, b in zip(G.transform(f), O.transform(f)))


def test_reference_vocabulary_subtree_fixture(pkg, ctx, oracle):
    """Real DBoW2 centres and idf weights (a part of the reference's vocabulary file, tests/golden/bow_refvoc_subtree.npz): ragged
    tree (2..8 children), stop words, five levels -- transform and query bit for bit."""
    from ssk_amd import capi
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bow_refvoc_subtree.npz"))
    v = {k[2:]: g[k] for k in g.files if k.startswith("v_")}
    G, O = both(pkg, ctx, oracle, v)
    gw, gx = G.transform(g["desc_in"])
    assert np.array_equal(gw, g["exp_word"]) and np.array_equal(gx, g["exp_val"])
    rng = np.random.default_rng(9)
    gdb, odb = capi.BowDb(ctx, G, 16, 512), oracle.bow_db(O.n_words)
    for f in bv.keyframe_sequence(v, rng, n_kf=12, n=400, noise=20):
        a, b = G.transform(f)
        oa, ob = O.transform(f)
        assert np.array_equal(a, oa) and np.array_equal(b, ob)
        gi, gs = gdb.query(a, b, 4)
        oi, os_ = odb.query(oa, ob, 4)
        assert np.array_equal(gi, oi) and np.array_equal(gs, os_)
        gdb.add(a, b); odb.add(oa, ob)


def test_query_without_bitmap_path(pkg, oracle):
    """Vocabularies beyond 512 Ki words have no room for the query bitmap in LDS: the score kernel then binary-searches the
    query.  SSK_BOW_NO_BITMAP=1 (read once per process) forces that path: run it in a process of its own."""
    import subprocess
    import sys
    code = """
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import __graft_entry__ as ge
import bow_vocab as bv, oracle_lib
pkg = ge.load_package()
from ssk_amd import capi
ctx = pkg.Context(0)
o = oracle_lib.load()
v = bv.synth(k=8, L=3, seed=51, ragged=True)
G, O = capi.BowVocab(ctx, v), o.bow_vocab(v)
gdb, odb = capi.BowDb(ctx, G, 64, 256), o.bow_db(O.n_words)
for f in bv.keyframe_sequence(v, np.random.default_rng(52), n_kf=30, n=200):
    w, x = G.transform(f)
    gi, gs = gdb.query(w, x, 4); oi, os_ = odb.query(w, x, 4)
    assert np.array_equal(gi, oi) and np.array_equal(gs, os_)
    gdb.add(w, x); odb.add(w, x)
print("OK")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, SSK_BOW_NO_BITMAP="1"), timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-2000:]
