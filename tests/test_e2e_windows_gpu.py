"""GPU: the stage-1 windows of define_new_keyframe (EST.cpp:782-784: the new edge optimised, the tracked landmarks held) assembled from a REAL
front-end + data-association run on the device -- the shape bench.py's `e2e` pass solves by the thousand: one edge, ~300 observations with a
one-step path, landmarks at the previous keyframe's triangulation, identity start -- checked per LM iteration against the oracle (the verdict of
round 4, 'weak' 1: such windows were only checked by 'error not increased')."""
import ctypes as C

import numpy as np
import pytest

import synth
import synth_lm as sl
from test_lm_gpu import check_against_oracle, to_capi

pytestmark = pytest.mark.gpu
TRACKED = 0                                                    # EST.h:102


def test_stage1_windows_from_a_frontend_and_da_run_per_iteration(pkg, ctx, oracle):
    import torch
    from ssk_amd import capi
    B, N = 8, 512
    cur = capi.KfBatch(ctx, B, synth.W, synth.H, N)
    prev = capi.KfBatch(ctx, B, synth.W, synth.H, N)
    for i in range(B):
        l, r = synth.synth_stereo_pair(40 + i)
        cur.upload(i, l, r)
    ctx.synchronize()
    ic, ip = cur.image_tensor(), prev.image_tensor()
    # the previous keyframe: the same scene from a camera displaced sideways -- image motion that differs from band to band (as bench.py does)
    for b0 in range(0, synth.H, 47):
        ip[:, :, b0:b0 + 47, :synth.W] = torch.roll(ic[:, :, b0:b0 + 47, :synth.W], shifts=-(2 + (b0 // 47) % 5), dims=-1)
    torch.cuda.synchronize()
    import bench as _b                                         # (the bench's parameter sets: what the e2e pass runs with)
    fpb, dab, lp, cp, fm = _b.params(capi, sl)
    prev.run_frontend(fpb); cur.run_frontend(fpb)
    dev = torch.device("cuda:0")
    d_idx = torch.empty(B * N, dtype=torch.int32, device=dev); d_dist = torch.empty(B * N, dtype=torch.int32, device=dev)
    d_st = torch.empty(B * N, dtype=torch.int32, device=dev); d_tr = torch.empty(B, dtype=torch.int32, device=dev)
    pairs = np.stack([np.arange(B), np.arange(B)], 1).astype(np.int32)
    ctx.da_run_batch(cur, prev, pairs, dab, d_idx, d_dist, d_st, d_tr)
    ctx.fm_run_batch(cur, prev, pairs, fm, d_idx, d_st, d_tr)
    ctx.cip_run_batch(cur, prev, pairs, cp, d_idx, d_st, d_tr)
    d_px = torch.zeros(B, N, 4, dtype=torch.float32, device=dev); d_pos = torch.zeros(B, N, 3, dtype=torch.float64, device=dev)
    cur.build_observations(lp, d_px, None)
    prev.build_observations(lp, None, d_pos)
    ctx.synchronize()
    _, nm = cur.counts()
    st = d_st.view(B, N).cpu().numpy(); ti = d_idx.view(B, N).cpu().numpy()
    px = d_px.cpu().numpy(); pos = d_pos.cpu().numpy()
    prm = sl.make_params()
    ident = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], np.float64)
    n_checked = 0
    for b in range(B):
        keep = np.flatnonzero((st[b] == TRACKED) & (ti[b] >= 0) & (np.arange(N) < nm[b]))
        assert len(keep) >= 100, (b, len(keep))
        m = len(keep)
        # one edge (slot 0, identity start), every tracked match an observation of its own landmark over that edge, landmarks held
        win = sl.Window(np.zeros(1, np.int32), ident, -np.ones(m, np.int32), pos[b][ti[b][keep]], np.arange(m, dtype=np.int32), px[b][keep],
                        np.arange(m + 1, dtype=np.int32), np.zeros(m, np.int32))
        res = check_against_oracle(ctx, oracle, win, prm)
        bb = capi.LmBatch(ctx, [win])
        assert bb.kernel_name() == "lm_pose_only_kernel"
        bb.close()
        assert res.err_final < res.err_init
        n_checked += 1
    assert n_checked == B
    cur.close(); prev.close()
