"""Regenerates the committed golden fixtures from the CPU oracle on seeded synthetic inputs.

The reference tree has no tests, fixtures or golden outputs and cannot be built here (SURVEY.md
section 4 / 8(c)), so these vectors pin the ORACLE against regressions; they are not reference outputs.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as ol   # noqa: E402
import synth              # noqa: E402


def make_da(oracle):
    for i, (seed, nf, nm, prm) in enumerate([(101, 160, 90, (1, 1, 370, 60.0)), (102, 500, 320, (1, 1, 370, 60.0))]):
        rng = np.random.default_rng(seed)
        a, b = synth.synth_kf_pair(rng, nf, nm)
        va, ka = ol.make_kf_view(**a)
        vb, kb = ol.make_kf_view(**b)
        out, st, tracked = oracle.data_association(va, vb, ol.DaParams(*prm))
        d = {"a_" + k: v for k, v in a.items()}
        d.update({"b_" + k: v for k, v in b.items()})
        np.savez_compressed(os.path.join(HERE, "da_case%d.npz" % i), prm=np.array(prm, np.float64),
                            exp_trainIdx=out["trainIdx"], exp_distance=out["distance"], exp_status=st,
                            exp_tracked=np.int32(tracked), **d)


def make_lm(oracle):
    import synth_lm as sl
    win, gt = sl.make_submap_problem(7, n_kf=14, n_lm=220, submap_size=5, obs_per_kf=60)
    out = oracle.lm_solve(win, sl.make_params())
    np.savez_compressed(os.path.join(HERE, "lm_case0.npz"), edge_unknown_slot=win.edge_unknown_slot,
                        edge_inv_pose=win.edge_inv_pose, lm_unknown_slot=win.lm_unknown_slot, lm_pos=win.lm_pos,
                        obs_lm=win.obs_lm, obs_px=win.obs_px, obs_path_off=win.obs_path_off, path_edge=win.path_edge,
                        iters=out["res"].iters, accepted=out["res"].accepted,
                        err_after=np.array([t.err_after for t in out["trace"]]),
                        final_poses=out["poses"], final_lms=out["lms"])


def make_cip(oracle):
    import synth_cip as sc
    c = sc.make_case(11, n_this=240, n_other=260, n_outliers=18)
    st, res, out = oracle.change_in_pose(c["this_obs"], c["other_obs"], c["train_idx"], c["status"], sc.identity_pose(), ol.kitti_cip_params())
    np.savez_compressed(os.path.join(HERE, "cip_case0.npz"), this_obs=c["this_obs"], other_obs=c["other_obs"], train_idx=c["train_idx"],
                        status=c["status"], exp_status=st, exp_residuals=res, exp_pose=np.array(out.pose[:]),
                        exp_iters=np.array([out.iters_initial, out.iters_final, out.n_inliers, out.n_rejected], np.int32))


def make_fmat(oracle):
    import synth_cip as sc
    c = sc.make_case(12, n_this=240, n_other=260, n_outliers=30)
    st, out = oracle.fundamental_filter(c["this_obs"][:, :2], c["other_obs"][:, :2], c["train_idx"], c["status"], ol.kitti_fm_params())
    np.savez_compressed(os.path.join(HERE, "fmat_case0.npz"), this_xy=c["this_obs"][:, :2].copy(), other_xy=c["other_obs"][:, :2].copy(),
                        train_idx=c["train_idx"], status=c["status"], exp_status=st, exp_F=np.array(out.F[:]),
                        exp_counts=np.array([out.n_inliers, out.hypotheses, out.best_hypothesis], np.int32))


def make_bow(oracle):
    import bow_vocab as bv
    v = bv.synth(k=8, L=3, seed=21)
    V = oracle.bow_vocab(v)
    D = oracle.bow_db(V.n_words)
    rng = np.random.default_rng(22)
    seq = bv.keyframe_sequence(v, rng, n_kf=10, n=150)
    seq.append(seq[2].copy())                        # a revisit: keyframe 10 sees what keyframe 2 saw
    desc_off, bow_off, bw, bx = [0], [0], [], []
    res_ids = np.zeros((len(seq), 4), np.int32); res_scores = np.zeros((len(seq), 4)); res_n = np.zeros(len(seq), np.int32)
    for e, f in enumerate(seq):
        w, x = V.transform(f)
        ids, sc = D.query(w, x, 4)                   # query before insertion, as EST.cpp:426 / :919
        res_n[e] = len(ids); res_ids[e, :len(ids)] = ids; res_scores[e, :len(ids)] = sc
        D.add(w, x)
        desc_off.append(desc_off[-1] + len(f)); bow_off.append(bow_off[-1] + len(w)); bw.append(w); bx.append(x)
    np.savez_compressed(os.path.join(HERE, "bow_case0.npz"), n_kf=np.int32(len(seq)), desc=np.concatenate(seq), desc_off=np.array(desc_off, np.int32),
                        bow_word=np.concatenate(bw), bow_val=np.concatenate(bx), bow_off=np.array(bow_off, np.int32), res_ids=res_ids,
                        res_scores=res_scores, res_n=res_n, **{"v_" + k: np.asarray(val) for k, val in v.items()})


def make_posegraph(oracle):
    import synth_pg as sp
    gph = sp.make_loop(31, n=48, edge_forward=False)
    X, r = oracle.pg_optimize(gph["init"], gph["edge_ij"], gph["edge_z"])
    np.savez_compressed(os.path.join(HERE, "posegraph_case0.npz"), init=gph["init"], edge_ij=gph["edge_ij"], edge_z=gph["edge_z"], exp_poses=X,
                        exp_counts=np.array([r.iters, r.accepted, r.stop_reason, r.pcg_iters_total], np.int32),
                        exp_cost=np.array([r.cost_init, r.cost_final]))


def make_bow_reference_subtree(oracle):
    """A part of the vocabulary the reference ships (demo/voc.yml.gz, data): the subtrees of the root's first two children with
    their real BRIEF centres and idf weights, words renumbered; plus oracle outputs on it.  Only where the file exists."""
    import bow_vocab as bv
    if not bv.have_reference_vocabulary():
        return
    v = bv.parse(bv.REF_VOC)
    keep = {1, 2}
    sel = []
    for i, (n, p) in enumerate(zip(v["node_id"], v["parent_id"])):          # file order: parents before children
        if int(n) in keep or int(p) in keep:
            keep.add(int(n))
            sel.append(i)
    sel = np.array(sel)
    old_words = {int(n): w for w, n in enumerate(v["word_node"])}
    word_node = np.array([int(v["node_id"][i]) for i in sel if int(v["node_id"][i]) in old_words], np.int32)
    sub = dict(k=v["k"], L=v["L"], scoring=v["scoring"], weighting=v["weighting"], node_id=v["node_id"][sel], parent_id=v["parent_id"][sel],
               weight=v["weight"][sel], desc=v["desc"][sel], word_node=word_node)
    V = oracle.bow_vocab(sub)
    rng = np.random.default_rng(41)
    f = bv.keyframe_descriptors(sub, rng, n=400, noise=20)
    w, x = V.transform(f)
    np.savez_compressed(os.path.join(HERE, "bow_refvoc_subtree.npz"), desc_in=f, exp_word=w, exp_val=x, **{"v_" + k: np.asarray(val) for k, val in sub.items()})


if __name__ == "__main__":
    o = ol.load()
    make_da(o)
    make_lm(o)
    make_cip(o)
    make_fmat(o)
    make_bow(o)
    make_posegraph(o)
    make_bow_reference_subtree(o)
    print("golden fixtures written to", HERE)
