"""ctypes binding of the CPU oracle (oracle/, test infrastructure).  Only tests/, smoke() and the
cpu_baseline leg of bench.py import this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class KfView(C.Structure):
    _fields_ = [("kp_left", C.c_void_p), ("n_left", C.c_int32),
                ("kp_right", C.c_void_p), ("n_right", C.c_int32),
                ("desc_left", C.c_void_p), ("desc_right", C.c_void_p),
                ("matches", C.c_void_p), ("n_matches", C.c_int32)]


class DaParams(C.Structure):
    _fields_ = [("filter_by_direction", C.c_int32), ("filter_by_orb_distance", C.c_int32),
                ("rows_offset", C.c_int32), ("max_orb_distance_da", C.c_float)]


class FrontendParams(C.Structure):
    _fields_ = [("n_feats", C.c_int32), ("fast_threshold", C.c_int32), ("n_levels", C.c_int32),
                ("nms_radius", C.c_int32), ("orb_max_distance", C.c_int32), ("max_y_diff", C.c_float),
                ("min_disparity", C.c_float), ("max_disparity", C.c_float), ("harris_k", C.c_float)]


class KfOut(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("kp_left", C.c_void_p), ("kp_right", C.c_void_p),
                ("desc_left", C.c_void_p), ("desc_right", C.c_void_p), ("matches", C.c_void_p),
                ("n_left", C.c_int32), ("n_right", C.c_int32), ("n_matches", C.c_int32)]


def alloc_kf_out(cls, cap):
    """(struct, arrays) with numpy-backed output arrays of `cap` records."""
    arrs = dict(kp_left=np.zeros(cap, KEYPOINT_DTYPE), kp_right=np.zeros(cap, KEYPOINT_DTYPE),
                desc_left=np.zeros((cap, 32), np.uint8), desc_right=np.zeros((cap, 32), np.uint8),
                matches=np.zeros(cap, DMATCH_DTYPE))
    o = cls(cap, arrs["kp_left"].ctypes.data, arrs["kp_right"].ctypes.data, arrs["desc_left"].ctypes.data,
            arrs["desc_right"].ctypes.data, arrs["matches"].ctypes.data, 0, 0, 0)
    return o, arrs


def trim_kf_out(o, arrs):
    return dict(kp_left=arrs["kp_left"][:o.n_left].copy(), kp_right=arrs["kp_right"][:o.n_right].copy(),
                desc_left=arrs["desc_left"][:o.n_left].copy(), desc_right=arrs["desc_right"][:o.n_right].copy(),
                matches=arrs["matches"][:o.n_matches].copy())


class CipParams(C.Structure):
    """ssk_cip_params (include/ssk_capi.h); defaults = [LEAST_SQUARES] / residual_th of demo/config_imgdir_kitti_srba.ini."""
    _fields_ = [(n, C.c_double) for n in ("fx_l", "fy_l", "cx_l", "cy_l", "fx_r", "fy_r", "cx_r", "cy_r", "baseline")] + [
        ("initial_max_iters", C.c_int32), ("max_iters", C.c_int32), ("max_incr_cost", C.c_int32), ("use_robust_kernel", C.c_int32),
        ("residual_threshold", C.c_double), ("min_mod_out_vector", C.c_double), ("kernel_param", C.c_double),
        ("bad_tracking_th", C.c_int32), ("reserved", C.c_int32), ("residual_th", C.c_double)]


class CipResult(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("cost", C.c_double), ("valid", C.c_int32), ("iters_initial", C.c_int32),
                ("iters_final", C.c_int32), ("n_pairs", C.c_int32), ("n_inliers", C.c_int32), ("n_rejected", C.c_int32)]


def kitti_cip_params(**over):
    import synth
    d = dict(synth.KITTI_CIP)
    d.update(over)
    return CipParams(**d)


class FmParams(C.Structure):
    """ssk_fm_params: fundamental-matrix filter (max_y_diff_epipolar / ransac_fit_prob of the reference ini)."""
    _fields_ = [("threshold", C.c_double), ("confidence", C.c_double), ("norm_cx", C.c_double), ("norm_cy", C.c_double),
                ("norm_scale", C.c_double), ("seed", C.c_uint64), ("max_iters", C.c_int32), ("min_points", C.c_int32)]


class FmResult(C.Structure):
    _fields_ = [("F", C.c_double * 9), ("valid", C.c_int32), ("n_pairs", C.c_int32), ("n_inliers", C.c_int32),
                ("n_rejected", C.c_int32), ("hypotheses", C.c_int32), ("best_hypothesis", C.c_int32)]


def kitti_fm_params(**over):
    import synth
    d = dict(synth.KITTI_FM)
    d.update(over)
    return FmParams(**d)


class PgParams(C.Structure):
    """ssk_pg_params; defaults: MRPT's 100 iterations / tau 1e-3, CG to 1e-10."""
    _fields_ = [("max_iters", C.c_int32), ("pcg_max_iters", C.c_int32), ("tau", C.c_double), ("max_lambda", C.c_double),
                ("converged_rel_decrease", C.c_double), ("pcg_tol", C.c_double)]

    def __init__(self, max_iters=100, pcg_max_iters=400, tau=1e-3, max_lambda=1e20, converged_rel_decrease=1e-9, pcg_tol=1e-10):
        super().__init__(max_iters, pcg_max_iters, tau, max_lambda, converged_rel_decrease, pcg_tol)


class PgResult(C.Structure):
    _fields_ = [("iters", C.c_int32), ("accepted", C.c_int32), ("stop_reason", C.c_int32), ("pcg_iters_total", C.c_int32),
                ("cost_init", C.c_double), ("cost_final", C.c_double), ("lambda_final", C.c_double)]


class StereoCam(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("fx_l", "fy_l", "cx_l", "cy_l", "fx_r", "fy_r", "cx_r", "cy_r", "baseline")]


def _p(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


class Oracle:
    def __init__(self, path):
        self.lib = C.CDLL(path)

    def hamming_match(self, q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros(len(q), DMATCH_DTYPE)
        n = self.lib.ssko_hamming_match(_p(q), len(q), _p(t), len(t), _p(out))
        return out[:n]

    def da_filter_direction(self, matches, this_xy, other_xy, offset, status):
        m = np.ascontiguousarray(matches, DMATCH_DTYPE)
        st = np.ascontiguousarray(status, np.int32).copy()
        a = np.ascontiguousarray(this_xy, np.float32)
        b = np.ascontiguousarray(other_xy, np.float32)
        self.lib.ssko_da_filter_direction(_p(m), len(m), _p(a), _p(b), int(offset), _p(st))
        return st

    def da_filter_orb_distance(self, matches, n_other, max_orb, status):
        m = np.ascontiguousarray(matches, DMATCH_DTYPE)
        st = np.ascontiguousarray(status, np.int32).copy()
        self.lib.ssko_da_filter_orb_distance.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        self.lib.ssko_da_filter_orb_distance(_p(m), len(m), int(n_other), float(max_orb), _p(st))
        return st

    def da_filters(self, matches, n_other, this_xy, other_xy, prm):
        st = np.zeros(len(matches), np.int32)
        if prm.filter_by_direction:
            st = self.da_filter_direction(matches, this_xy, other_xy, prm.rows_offset, st)
        if prm.filter_by_orb_distance:
            st = self.da_filter_orb_distance(matches, n_other, prm.max_orb_distance_da, st)
        return st, int((st == 0).sum())

    def data_association(self, this_view, other_view, prm):
        n = this_view.n_matches
        out = np.zeros(n, DMATCH_DTYPE)
        st = np.zeros(n, np.int32)
        tracked = self.lib.ssko_data_association(C.byref(this_view), C.byref(other_view), C.byref(prm), _p(out), _p(st))
        return out, st, tracked

    def propagate_ids(self, tracking_first, other_ids, last_id):
        tf = np.ascontiguousarray(tracking_first, np.int32)
        n_sim, n_m = tf.shape
        ids = [np.ascontiguousarray(o, np.uint64) for o in other_ids]
        arr = (C.c_void_p * max(1, n_sim))(*[o.ctypes.data for o in ids])
        last = C.c_uint64(last_id)
        out = np.zeros(n_m, np.uint64)
        per = np.zeros(max(1, n_sim), np.int32)
        nt = self.lib.ssko_propagate_ids(n_m, n_sim, _p(tf), arr, C.byref(last), _p(out), _p(per))
        return out, per[:n_sim], nt, last.value

    # ---- front-end -----------------------------------------------------------------------------
    def fast_score_map(self, img, threshold):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        sc = np.zeros((h, w), np.uint8)
        self.lib.ssko_fast_score_map(_p(img), w, h, w, int(threshold), _p(sc))
        return sc

    def fast_candidates(self, img, threshold):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        keys = np.zeros((w // 2 + 1) * (h // 2 + 1), np.uint32)
        n = self.lib.ssko_fast_candidates(_p(img), w, h, w, int(threshold), _p(keys), len(keys))
        return keys[:n].copy()

    def set_orb_steering(self, mode):
        """C7 of the front-end contract: 1 (default) = cv::ORB's cos / sin of the angle, 0 = exact centroid direction (rounds 1-2)."""
        self.lib.ssko_set_orb_steering(int(mode))

    def orb_detect(self, img, prm, pattern):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = prm.n_feats
        kp = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        pat = np.ascontiguousarray(pattern, np.int8)
        n = self.lib.ssko_orb_detect(_p(img), w, h, w, C.byref(prm), _p(pat), _p(kp), _p(desc), cap)
        assert n >= 0
        return kp[:n].copy(), desc[:n].copy()

    def stereo_match(self, kl, dl, kr, dr, prm):
        kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE)
        kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
        dl = np.ascontiguousarray(dl, np.uint8)
        dr = np.ascontiguousarray(dr, np.uint8)
        out = np.zeros(max(1, len(kl)), DMATCH_DTYPE)
        n = self.lib.ssko_stereo_match(_p(kl), len(kl), _p(dl), _p(kr), len(kr), _p(dr), C.byref(prm), _p(out))
        return out[:n].copy()

    def frontend(self, left, right, prm, pattern):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        h, w = left.shape
        pat = np.ascontiguousarray(pattern, np.int8)
        o, arrs = alloc_kf_out(KfOut, prm.n_feats)
        rc = self.lib.ssko_frontend(_p(left), _p(right), w, h, w, C.byref(prm), _p(pat), C.byref(o))
        assert rc == 0
        return trim_kf_out(o, arrs)

    # ---- relative bundle adjustment -----------------------------------------------------------------
    def lm_total_error(self, win, prm, poses=None, lms=None):
        poses = np.ascontiguousarray(win.edge_inv_pose if poses is None else poses, np.float64)
        lms = np.ascontiguousarray(win.lm_pos if lms is None else lms, np.float64)
        wc = win.c()
        self.lib.ssko_lm_total_error.restype = C.c_double
        return self.lib.ssko_lm_total_error(C.byref(wc), C.byref(prm), _p(poses), _p(lms))

    def lm_gradient(self, win, prm, poses=None, lms=None):
        poses = np.ascontiguousarray(win.edge_inv_pose if poses is None else poses, np.float64)
        lms = np.ascontiguousarray(win.lm_pos if lms is None else lms, np.float64)
        gp = np.zeros(6 * win.E + 1)
        gf = np.zeros(3 * win.L + 1)
        wc = win.c()
        self.lib.ssko_lm_gradient(C.byref(wc), C.byref(prm), _p(poses), _p(lms), _p(gp), _p(gf))
        return gp[:6 * win.E], gf[:3 * win.L]

    def se3_left_update(self, eps, T):
        out = np.zeros(12)
        e = np.ascontiguousarray(eps, np.float64)
        t = np.ascontiguousarray(T, np.float64)
        self.lib.ssko_se3_left_update(_p(e), _p(t), _p(out))
        return out

    def lm_solve(self, win, prm, trace=True):
        """Returns dict(poses, lms, result, trace entries, dp, df)."""
        import synth_lm
        poses = win.edge_inv_pose.copy()
        lms = win.lm_pos.copy()
        res = synth_lm.LmResultC()
        n = prm.max_iters
        tr = (synth_lm.LmTraceC * n)()
        dp = np.zeros((n, 6 * win.E + 1))
        df = np.zeros((n, 3 * win.L + 1))
        dpc = np.zeros(n * 6 * win.E + 1)
        dfc = np.zeros(n * 3 * win.L + 1)
        wc = win.c()
        self.lib.ssko_lm_solve(C.byref(wc), C.byref(prm), _p(poses), _p(lms), C.byref(res),
                               tr if trace else None, _p(dpc) if trace else None, _p(dfc) if trace else None)
        dp = dpc[:n * 6 * win.E].reshape(n, 6 * win.E) if win.E else np.zeros((n, 0))
        df = dfc[:n * 3 * win.L].reshape(n, 3 * win.L) if win.L else np.zeros((n, 0))
        return dict(poses=poses, lms=lms, res=res, trace=list(tr)[:res.iters], dp=dp[:res.iters], df=df[:res.iters])

    def change_in_pose(self, this_obs, other_obs, train_idx, status, init_pose, prm):
        """Returns (status after the filter, residuals of the tracked pairs, CipResult)."""
        to = np.ascontiguousarray(this_obs, np.float32).reshape(-1, 4)
        oo = np.ascontiguousarray(other_obs, np.float32).reshape(-1, 4)
        ti = np.ascontiguousarray(train_idx, np.int32)
        st = np.ascontiguousarray(status, np.int32).copy()
        T = np.ascontiguousarray(init_pose, np.float64).reshape(12)
        res = np.zeros(len(to) + 1)
        out = CipResult()
        self.lib.ssko_change_in_pose.restype = C.c_int
        self.lib.ssko_change_in_pose(_p(to), len(to), _p(oo), len(oo), _p(ti), _p(st), _p(T), C.byref(prm), _p(res), C.byref(out))
        return st, res[:out.n_pairs].copy(), out

    def fundamental_filter(self, this_xy, other_xy, train_idx, status, prm):
        """Returns (status after the filter, FmResult)."""
        tx = np.ascontiguousarray(this_xy, np.float32).reshape(-1, 2)
        ox = np.ascontiguousarray(other_xy, np.float32).reshape(-1, 2)
        ti = np.ascontiguousarray(train_idx, np.int32)
        st = np.ascontiguousarray(status, np.int32).copy()
        out = FmResult()
        self.lib.ssko_fundamental_filter.restype = C.c_int
        self.lib.ssko_fundamental_filter(_p(tx), len(tx), _p(ox), len(ox), _p(ti), _p(st), C.byref(prm), C.byref(out))
        return st, out

    # ---- bag of words (oracle_bow.c) ----
    def bow_vocab(self, v):
        """v: dict from bow_vocab.parse / synth -> OracleBowVocab."""
        return OracleBowVocab(self.lib, v)

    def bow_db(self, n_words):
        return OracleBowDb(self.lib, n_words)

    # ---- global pose graph (oracle_posegraph.c) ----
    def pg_cost(self, X, edge_ij, edge_z):
        X = np.ascontiguousarray(X, np.float64).reshape(-1, 12)
        ij = np.ascontiguousarray(edge_ij, np.int32).reshape(-1, 2)
        z = np.ascontiguousarray(edge_z, np.float64).reshape(-1, 12)
        self.lib.ssko_pg_cost.restype = C.c_double
        return self.lib.ssko_pg_cost(len(ij), _p(ij), _p(z), _p(X))

    def pg_optimize(self, X, edge_ij, edge_z, fixed=0, prm=None):
        """Returns (optimised poses [n, 12], PgResult)."""
        X = np.ascontiguousarray(X, np.float64).reshape(-1, 12).copy()
        ij = np.ascontiguousarray(edge_ij, np.int32).reshape(-1, 2)
        z = np.ascontiguousarray(edge_z, np.float64).reshape(-1, 12)
        prm = prm or PgParams()
        out = PgResult()
        self.lib.ssko_pg_optimize(len(X), _p(X), len(ij), _p(ij), _p(z), int(fixed), C.byref(prm), C.byref(out))
        return X, out

    def so3_log(self, T):
        T = np.ascontiguousarray(T, np.float64).reshape(12)
        w = np.zeros(3)
        self.lib.ssko_so3_log(_p(T), _p(w))
        return w

    def project_match_to_3d(self, ul, vl, ur, cam):
        out = (C.c_double * 3)()
        self.lib.ssko_project_match_to_3d.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        self.lib.ssko_project_match_to_3d(ul, vl, ur, C.byref(cam), out)
        return np.array(out[:])


class OracleBowVocab:
    def __init__(self, lib, v):
        self.lib = lib
        self.n_words = len(v["word_node"])
        nid = np.ascontiguousarray(v["node_id"], np.int32); pid = np.ascontiguousarray(v["parent_id"], np.int32)
        w = np.ascontiguousarray(v["weight"], np.float64); d = np.ascontiguousarray(v["desc"], np.uint8)
        wn = np.ascontiguousarray(v["word_node"], np.int32)
        lib.ssko_bow_vocab_create.restype = C.c_void_p
        self.h = lib.ssko_bow_vocab_create(int(v["k"]), int(v["L"]), len(nid), _p(nid), _p(pid), _p(w), _p(d), len(wn), _p(wn))
        if not self.h:
            raise ValueError("malformed vocabulary")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ssko_bow_vocab_destroy.argtypes = [C.c_void_p]
            self.lib.ssko_bow_vocab_destroy(self.h)
            self.h = None

    def word(self, feat):
        f = np.ascontiguousarray(feat, np.uint8).reshape(32)
        w = C.c_double()
        self.lib.ssko_bow_word.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        return self.lib.ssko_bow_word(self.h, _p(f), C.byref(w)), w.value

    def transform(self, feats):
        f = np.ascontiguousarray(feats, np.uint8).reshape(-1, 32)
        words = np.zeros(len(f) + 1, np.int32); vals = np.zeros(len(f) + 1)
        self.lib.ssko_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        n = self.lib.ssko_bow_transform(self.h, _p(f), len(f), _p(words), _p(vals))
        return words[:n].copy(), vals[:n].copy()


class OracleBowDb:
    def __init__(self, lib, n_words):
        self.lib = lib
        lib.ssko_bow_db_create.restype = C.c_void_p
        self.h = lib.ssko_bow_db_create(int(n_words))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ssko_bow_db_destroy.argtypes = [C.c_void_p]
            self.lib.ssko_bow_db_destroy(self.h)
            self.h = None

    def __len__(self):
        self.lib.ssko_bow_db_size.argtypes = [C.c_void_p]
        return self.lib.ssko_bow_db_size(self.h)

    def add(self, words, vals):
        w = np.ascontiguousarray(words, np.int32); v = np.ascontiguousarray(vals, np.float64)
        self.lib.ssko_bow_db_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        return self.lib.ssko_bow_db_add(self.h, _p(w), _p(v), len(w))

    def query(self, words, vals, max_results=4, max_id=-1):
        w = np.ascontiguousarray(words, np.int32); v = np.ascontiguousarray(vals, np.float64)
        cap = max_results if max_results > 0 else max(len(self), 1)
        ids = np.zeros(cap, np.int32); sc = np.zeros(cap)
        self.lib.ssko_bow_db_query.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        n = self.lib.ssko_bow_db_query(self.h, _p(w), _p(v), len(w), int(max_results), int(max_id), _p(ids), _p(sc))
        return ids[:n].copy(), sc[:n].copy()


class OracleEdgePolicy:
    """oracle/oracle_edge_policy.c: mySRBA::edge_creation_policy (SRBA.h:144-308) and its local-map bookkeeping."""

    def __init__(self, lib, submap_size, min_obs_lc, max_tree_depth, max_optimize_depth):
        self.lib = lib
        lib.ssko_ecp_create.restype = C.c_void_p
        lib.ssko_ecp_create.argtypes = [C.c_int] * 4
        for f in ("ssko_ecp_destroy", "ssko_ecp_set_initial_kf_pose", "ssko_ecp_set_edge_pose", "ssko_ecp_num_edges", "ssko_ecp_get_edge",
                  "ssko_ecp_kf_centers", "ssko_ecp_localmap_kfs", "ssko_ecp_define_kf"):
            getattr(lib, f).argtypes = None
        self.h = C.c_void_p(lib.ssko_ecp_create(submap_size, min_obs_lc, max_tree_depth, max_optimize_depth))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ssko_ecp_destroy(self.h)
            self.h = None

    def set_initial_kf_pose(self, pose):
        p = np.ascontiguousarray(pose, np.float64).reshape(12)
        self.lib.ssko_ecp_set_initial_kf_pose(self.h, _p(p))

    def set_edge_pose(self, e, pose):
        p = np.ascontiguousarray(pose, np.float64).reshape(12)
        self.lib.ssko_ecp_set_edge_pose(self.h, C.c_int(e), _p(p))

    def define_kf(self, feat_ids):
        ids = np.ascontiguousarray(feat_ids, np.uint64)
        return self.lib.ssko_ecp_define_kf(self.h, _p(ids), C.c_int(len(ids)))

    def num_edges(self):
        return self.lib.ssko_ecp_num_edges(self.h)

    def edge(self, e):
        f, t = C.c_int(), C.c_int()
        p = np.zeros(12)
        assert self.lib.ssko_ecp_get_edge(self.h, C.c_int(e), C.byref(f), C.byref(t), _p(p)) == 0
        return f.value, t.value, p

    def kf_centers(self, kf):
        out = np.zeros(8, np.int32)
        n = self.lib.ssko_ecp_kf_centers(self.h, C.c_int(kf), _p(out))
        return [int(v) for v in out[:n]]

    def localmap_kfs(self, c):
        out = np.zeros(64, np.int32)
        n = self.lib.ssko_ecp_localmap_kfs(self.h, C.c_int(c), _p(out))
        return None if n < 0 else [int(v) for v in out[:n]]


def make_kf_view(kp_left, kp_right, desc_left, desc_right, matches):
    kp_left = np.ascontiguousarray(kp_left, dtype=KEYPOINT_DTYPE)
    kp_right = np.ascontiguousarray(kp_right, dtype=KEYPOINT_DTYPE)
    desc_left = np.ascontiguousarray(desc_left, dtype=np.uint8).reshape(-1, 32)
    desc_right = np.ascontiguousarray(desc_right, dtype=np.uint8).reshape(-1, 32)
    matches = np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)
    v = KfView(_p(kp_left), len(kp_left), _p(kp_right), len(kp_right), _p(desc_left), _p(desc_right),
               _p(matches), len(matches))
    return v, (kp_left, kp_right, desc_left, desc_right, matches)


def default_pattern():
    """The default BRIEF pattern, parsed from include/ssk_brief_pattern.h (data, not code)."""
    import re
    src = open(os.path.join(ROOT, "include", "ssk_brief_pattern.h")).read()
    body = src[src.index("{") + 1:src.rindex("}")]
    v = np.array([int(t) for t in re.findall(r"-?\d+", body)], np.int8)
    assert v.size == 1024
    return v


_ORACLE = None


def load(march=None, out=None, opt=None):
    global _ORACLE
    if march or out or opt:
        import __graft_entry__ as ge
        return Oracle(ge.build_oracle(march=march, out=out, opt=opt))
    if _ORACLE is None:
        path = os.path.join(ROOT, "oracle", "_build", "libssk_oracle.so")
        srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
                if f.endswith((".c", ".h"))]
        if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(s) for s in srcs):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        _ORACLE = Oracle(path)
    return _ORACLE
