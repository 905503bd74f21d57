"""GPU: the shared-landmark exchange of BASELINE configs[4] with its real meaning.  Two config-3-shaped submaps observe 24
landmarks of one identity each (tests/synth_lm.make_shared_submaps); neither window optimises them.  One exchange round = both
windows' LM solves, ssk_lm_batch_pack_shared (every window's g / H blocks of the shared landmarks at its solved state), the all-gather
(here: the two windows of one batch laid out rank-major), ssk_lm_stitch_shared (rank-ordered sums, Gauss-Newton step of the landmarks),
ssk_lm_batch_set_shared.  Checked:
  * the packed blocks and the update against the numpy restatement tests/shared_ref.py,
  * the rounds converge -- block-coordinate descent on the joint problem -- to the JOINT single-process solve of the union (the
    CPU oracle on tests/synth_lm.join_windows): edge poses, private and shared landmarks within 1e-6."""
import ctypes as C

import numpy as np
import pytest

import shared_ref as sr
import synth_lm as sl

pytestmark = pytest.mark.gpu


def to_capi(prm):
    from ssk_amd import capi
    out = capi.LmParams()
    C.memmove(C.byref(out), C.byref(prm), C.sizeof(out))
    return out


def test_exchange_rounds_converge_to_the_joint_solve(ctx, oracle):
    import torch
    from ssk_amd import capi
    S = 24
    wins, idxs, sh = sl.make_shared_submaps(n_sub=2, n_shared=S, seed0=50, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60, opt_depth=2)
    assert all(int((i >= 0).sum()) == S for i in idxs) and all(w.E >= 3 for w in wins)
    prm = sl.make_params(max_iters=50, conv=1e-14)
    joint, e_off, lm_map, sj = sl.join_windows(wins, idxs)
    J = oracle.lm_solve(joint, prm, trace=False)
    assert J["res"].err_final < 0.02 * J["res"].err_init
    dev = torch.device("cuda:0")
    b = capi.LmBatch(ctx, wins)
    cp = to_capi(prm)
    d_idx = [torch.from_numpy(i.astype(np.int32)).to(dev) for i in idxs]
    gathered = torch.zeros(2 * S * 12, dtype=torch.float64, device=dev)
    pos = torch.from_numpy(sh["init"].reshape(-1).copy()).to(dev)
    errs = []
    for rnd in range(70):
        b.run(cp)                                           # (continues from the state of the previous round)
        res = b.results()
        errs.append(sum(r.err_final for r in res))
        for r in range(2):
            b.pack_shared(r, d_idx[r], cp, gathered[r * S * 12:(r + 1) * S * 12])
        if rnd == 0:                                        # the device blocks and the update against the numpy restatement
            ctx.synchronize()
            g = gathered.cpu().numpy().reshape(2, S, 12)
            for r in range(2):
                p_r, l_r = b.state(r)
                exp = sr.blocks(wins[r], prm, p_r, l_r, idxs[r])
                assert np.allclose(g[r], exp, rtol=1e-9, atol=1e-9 * np.abs(exp).max()), r
                assert np.all(g[r][:, 6] > 0) and np.all(g[r][:, 9] > 0) and np.all(g[r][:, 11] > 0)
            exp_pos = sr.consensus(g, sh["init"])
        ctx.stitch_shared(gathered, 2, S, pos)
        if rnd == 0:
            ctx.synchronize()
            assert np.allclose(pos.cpu().numpy().reshape(S, 3), exp_pos, rtol=1e-10, atol=1e-12)
        for r in range(2):
            b.set_shared(r, d_idx[r], pos)
    ctx.synchronize()
    assert errs[-1] < errs[0] and abs(errs[-1] - J["res"].err_final) <= 1e-9 * J["res"].err_final, (errs[0], errs[-1], J["res"].err_final)
    shared = pos.cpu().numpy().reshape(S, 3)
    assert np.abs(shared - J["lms"][sj]).max() <= 1e-6
    for r in range(2):
        p_r, l_r = b.state(r)
        assert np.abs(p_r - J["poses"][e_off[r]:e_off[r] + len(p_r)]).max() <= 1e-6, r
        ref = J["lms"][lm_map[r]]                           # 1e-6 RELATIVE: a landmark 40 m away is known to its depth's share of that
        assert (np.abs(l_r - ref) <= 1e-6 * np.maximum(1.0, np.linalg.norm(ref, axis=1))[:, None]).all(), r
        assert np.array_equal(l_r[idxs[r]], shared)         # every window holds the consensus positions
    # ssk_lm_batch_reset keeps the consensus positions (they are part of the initial state now)
    b.reset()
    assert np.array_equal(b.state(0)[1][idxs[0]], shared)
    b.close()


def test_configs4_at_its_stated_size_accelerated_rounds(ctx, oracle, pkg):
    """BASELINE configs[4] at its stated size: 8 config-3-shaped submaps (200 keyframes / 5 000 landmarks, window = the 10 most recent
    edges, seeds 50..57) + 256 landmarks of one identity each, every one observed from two of the eight submaps.  The 8 windows of one
    batch stand for the 8 ranks (rank-major gathered buffer); the rounds are mixed by SharedLandmarkAccelerator (Anderson, m = 5) and
    stop when the consensus moves by less than 1e-8; the result is the oracle's JOINT solve of the union -- 80 edges, ~3 000 landmarks --
    within 1e-6, in a fraction of the 70 plain rounds."""
    import torch
    from ssk_amd import capi
    n_sub, S = 8, 256
    wins, idxs, sh = sl.make_shared_submaps(n_sub=n_sub, n_shared=S, seed0=50, n_kf=200, n_lm=5000, submap_size=10, max_tree_depth=4, opt_depth=2,
                                            obs_per_kf=250, lm_fwd=(3, 20), z_vis=(2.5, 22))
    assert all(w.E == 10 for w in wins) and all(int((i >= 0).sum()) >= 16 for i in idxs)
    prm = sl.make_params(max_iters=50, conv=1e-14)
    joint, e_off, lm_map, sj = sl.join_windows(wins, idxs)
    seen = sj >= 0                                          # (a shared landmark no submap happens to see stays where it started)
    J = oracle.lm_solve(joint, prm, trace=False)
    dev = torch.device("cuda:0")
    b = capi.LmBatch(ctx, wins)
    assert b.kernel_name() == "lm_split_linearise_kernel"   # (a batch of 8 windows of 1.4 k observations: the phase-split kernels; 9+ windows: the node kernel)
    cp = to_capi(prm)
    d_idx = [torch.from_numpy(i.astype(np.int32)).to(dev) for i in idxs]
    gathered = torch.zeros(n_sub * S * 12, dtype=torch.float64, device=dev)
    pos = torch.from_numpy(sh["init"].reshape(-1).copy()).to(dev)
    acc = pkg.dist.SharedLandmarkAccelerator(m=5)
    rounds = 0
    for rnd in range(40):
        b.run(cp); b.results()
        for r in range(n_sub):
            b.pack_shared(r, d_idx[r], cp, gathered[r * S * 12:(r + 1) * S * 12])
        x = pos.cpu().numpy().copy()
        ctx.stitch_shared(gathered, n_sub, S, pos)
        ctx.synchronize()
        nxt = acc.step(x, pos.cpu().numpy())
        pos.copy_(torch.from_numpy(nxt))
        for r in range(n_sub):
            b.set_shared(r, d_idx[r], pos)
        rounds = rnd + 1
        if np.abs(nxt - x).max() < 1e-8:
            break
    ctx.synchronize()
    shared = pos.cpu().numpy().reshape(S, 3)
    assert rounds <= 25, rounds
    assert np.abs(shared[seen] - J["lms"][sj[seen]]).max() <= 1e-6, (rounds, np.abs(shared[seen] - J["lms"][sj[seen]]).max())
    b.run(cp); b.results()                                  # the windows at the final consensus
    for r in range(n_sub):
        p_r, l_r = b.state(r)
        assert np.abs(p_r - J["poses"][e_off[r]:e_off[r] + len(p_r)]).max() <= 1e-6, r
    print("configs[4] at size: %d accelerated rounds to a consensus step below 1e-8" % rounds)
    b.close()


def test_pack_of_a_landmark_nobody_observes_and_bad_arguments(ctx):
    import torch
    from ssk_amd import capi
    wins, idxs, sh = sl.make_shared_submaps(n_sub=2, n_shared=8, seed0=60, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60, opt_depth=2)
    b = capi.LmBatch(ctx, wins[:1])
    dev = torch.device("cuda:0")
    idx = idxs[0].astype(np.int32).copy()
    idx[3] = -1                                             # not in this window: a zero block, and the stitch leaves it where it is
    d_idx = torch.from_numpy(idx).to(dev)
    out = torch.full((8 * 12,), 7.0, dtype=torch.float64, device=dev)
    b.pack_shared(0, d_idx, to_capi(sl.make_params()), out)
    pos = torch.from_numpy(sh["init"].reshape(-1).copy()).to(dev)
    ctx.stitch_shared(out, 1, 8, pos)
    ctx.synchronize()
    o = out.cpu().numpy().reshape(8, 12)
    assert np.all(o[3] == 0) and np.all(o[[0, 1, 2, 4, 5, 6, 7], 6] > 0)
    p = pos.cpu().numpy().reshape(8, 3)
    assert np.array_equal(p[3], sh["init"][3]) and not np.array_equal(p[0], sh["init"][0])
    with pytest.raises(capi.SskError):
        b.pack_shared(1, d_idx, to_capi(sl.make_params()), out)       # no such window
    b.close()


def test_pack_from_the_node_tables_equals_the_pack_from_the_raw_arrays(ctx, monkeypatch):
    """Round 6: a lean batch keeps no raw observation arrays for its node / phase-split windows; ssk_lm_batch_pack_shared reads such a window's own
    tables (pack_shared_node_kernel: the suffix trie unrolled into the path, the observations in table order).  Against the pack of the same window
    in a batch of the full builder (SSK_LM_KERNEL=tiled: raw arrays, pack_shared_kernel): positions equal, g / H equal to rounding."""
    import torch
    from ssk_amd import capi
    wins, idxs, sh = sl.make_shared_submaps(n_sub=2, n_shared=32, seed0=70, n_kf=40, n_lm=900, submap_size=6, obs_per_kf=120, opt_depth=2)
    dev = torch.device("cuda:0")
    d_idx = torch.from_numpy(idxs[0].astype(np.int32)).to(dev)
    outs = []
    for kernel in (None, "tiled"):
        if kernel:
            monkeypatch.setenv("SSK_LM_KERNEL", kernel)
        else:
            monkeypatch.delenv("SSK_LM_KERNEL", raising=False)
        b = capi.LmBatch(ctx, wins[:1])
        out = torch.zeros(32 * 12, dtype=torch.float64, device=dev)
        b.pack_shared(0, d_idx, to_capi(sl.make_params()), out)
        ctx.synchronize()
        outs.append((b.kernel_name(), out.cpu().numpy().reshape(32, 12)))
        b.close()
    monkeypatch.delenv("SSK_LM_KERNEL", raising=False)
    (n0, a), (n1, c) = outs
    assert n0.startswith("lm_solve_node") or n0.startswith("lm_split"), n0
    assert "tiled" in n1 or "coop" in n1, n1
    assert np.array_equal(a[:, :3], c[:, :3]) and np.abs(a[:, 6]).min() > 0
    assert np.allclose(a[:, 3:], c[:, 3:], rtol=1e-11, atol=1e-9 * np.abs(c[:, 3:]).max())
