"""CPU: ssk_rba_obs_from_da_batch (include/ssk_rba.h) -- the observation lists of a batch of streams' new keyframes from their stereo matches and the data
association with the previous keyframe (EST.cpp:666-704 per stream: a tracked match inherits the landmark id of the previous keyframe's match, every other
match founds a new landmark).  A host function of libssk_facade.so: no GPU call is made (the library only has to load)."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def capi():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from ssk_amd import capi as c
    try:
        c.load_facade_library()
    except OSError as e:                       # (built by __graft_entry__.build(); a tree without it cannot run this test)
        pytest.skip("libssk_facade.so not built: %s" % e)
    return c


def reference(n_matches, status, train_idx, ids_prev, next_id, cap):
    """plain Python restatement of the id rule"""
    B = len(n_matches)
    ids = -np.ones((B, cap), np.int64)
    nxt = next_id.copy()
    for b in range(B):
        for i in range(min(n_matches[b], cap)):
            inherit = -1
            if ids_prev is not None and status[b, i] == 0 and 0 <= train_idx[b, i] < cap:
                inherit = ids_prev[b, train_idx[b, i]]
            if inherit < 0:
                inherit = (b << 40) + nxt[b]; nxt[b] += 1
            ids[b, i] = inherit
    return ids, nxt


def test_obs_from_da_batch_ids_and_records(capi):
    lib = capi.load_facade_library()
    rng = np.random.default_rng(2)
    B, cap = 5, 32
    nm = np.array([32, 17, 0, 40, 9], np.int32)                       # (40 > cap: clamped)
    px = rng.uniform(0, 1200, (B, cap, 4)).astype(np.float32)
    pos = rng.normal(0, 10, (B, cap, 3))
    next_id = np.array([0, 5, 7, 0, 100], np.int64)
    obs = np.zeros((B, cap), capi.RBA_OBS_DTYPE)
    ids0 = np.empty((B, cap), np.int64)
    n0 = next_id.copy()
    assert lib.ssk_rba_obs_from_da_batch(B, cap, capi._ptr(nm), None, None, capi._ptr(px), capi._ptr(pos), None, capi._ptr(n0), capi._ptr(ids0), capi._ptr(obs)) == 0
    exp0, nx0 = reference(nm, None, None, None, next_id, cap)
    assert np.array_equal(ids0, exp0) and np.array_equal(n0, nx0)
    for b in range(B):
        k = min(nm[b], cap)
        assert np.array_equal(obs["feat_id"][b, :k].astype(np.int64), exp0[b, :k])
        assert np.array_equal(np.stack([obs[f][b, :k] for f in ("left_x", "left_y", "right_x", "right_y")], 1), px[b, :k])
        assert np.array_equal(obs["rel_pos"][b, :k], pos[b, :k])
        assert (b << 40) <= exp0[b, :k].min(initial=b << 40) and (k == 0 or exp0[b, :k].max() < ((b + 1) << 40))      # ids of different streams never collide
    # second keyframe: some matches tracked (status 0 = SSK_DA_TRACKED), some rejected, some pointing beyond the previous keyframe's matches (-> new)
    status = rng.integers(0, 7, (B, cap)).astype(np.int32)
    train = rng.integers(-1, cap + 3, (B, cap)).astype(np.int32)
    ids1 = np.empty((B, cap), np.int64)
    n1 = n0.copy()
    assert lib.ssk_rba_obs_from_da_batch(B, cap, capi._ptr(nm), capi._ptr(status), capi._ptr(train), capi._ptr(px), capi._ptr(pos), capi._ptr(ids0), capi._ptr(n1),
                                         capi._ptr(ids1), capi._ptr(obs)) == 0
    exp1, nx1 = reference(nm, status, train, ids0, n0, cap)
    assert np.array_equal(ids1, exp1) and np.array_equal(n1, nx1)
    tracked = (status == 0) & (train >= 0) & (train < cap) & (np.arange(cap)[None, :] < np.minimum(nm, cap)[:, None])
    assert tracked.any() and (np.take_along_axis(ids0, np.clip(train, 0, cap - 1).astype(np.int64), 1)[tracked] >= 0).any()
    # bad arguments
    assert lib.ssk_rba_obs_from_da_batch(0, cap, capi._ptr(nm), None, None, capi._ptr(px), capi._ptr(pos), None, capi._ptr(n1), capi._ptr(ids1), capi._ptr(obs)) != 0
    assert lib.ssk_rba_obs_from_da_batch(B, cap, capi._ptr(nm), None, None, capi._ptr(px), capi._ptr(pos), capi._ptr(ids0), capi._ptr(n1), capi._ptr(ids1), capi._ptr(obs)) != 0
