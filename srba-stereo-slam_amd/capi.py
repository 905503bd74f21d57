"""ctypes binding of libssk_hip.so -- exactly the entry points declared in include/ssk_capi.h."""
import atexit
import ctypes as C
import os
import re
import sys
import time
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "ssk_capi.h")

# layout of cv::KeyPoint / cv::DMatch (reference: src/srba-stereo-slam_utils.h:47-48)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16


class SskError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("ssk error %d: %s" % (code, text))
        self.code = code


def lib_path():
    # SSK_LIB: another build of the same library (same-box A/B timing of kernel variants, tools/lm_ab.py); default: the in-tree build
    return os.environ.get("SSK_LIB") or os.path.join(HERE, "libssk_hip.so")


def declared_symbols():
    """Every function name declared in include/ssk_capi.h (used by the CPU export test)."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssk_[a-z0-9_]+)\s*\(", src)))


_LIB = None
_LIVE = weakref.WeakSet()     # contexts / batches still open: closed before interpreter (and HIP runtime) teardown


@atexit.register
def _close_all():
    objs = list(_LIVE)
    for o in objs:
        if not isinstance(o, Context):
            o.close()
    for o in objs:
        if isinstance(o, Context):
            o.close()


def load_library():
    """Load libssk_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ImportError("libssk_hip.so is not built: run `python __graft_entry__.py build` "
                              "(there is no CPU fallback)")
        # One HIP/HSA runtime per process: torch ships its own libamdhip64.so.7 / libhsa-runtime64 and
        # libssk_hip.so needs the same SONAME.  Importing torch first makes the dynamic loader bind our
        # library to the runtime torch already loaded (a second HSA runtime in the process cannot open
        # the GPU: "No HIP GPUs are available").  A C++ host without torch simply uses /opt/rocm's.
        import torch  # noqa: F401
        _LIB = C.CDLL(p)
        _LIB.ssk_last_error.restype = C.c_char_p
        _LIB.ssk_stream.restype = C.c_void_p
    return _LIB


class KfView(C.Structure):
    _fields_ = [("kp_left", C.c_void_p), ("n_left", C.c_int32),
                ("kp_right", C.c_void_p), ("n_right", C.c_int32),
                ("desc_left", C.c_void_p), ("desc_right", C.c_void_p),
                ("matches", C.c_void_p), ("n_matches", C.c_int32)]


class DaParams(C.Structure):
    _fields_ = [("filter_by_direction", C.c_int32), ("filter_by_orb_distance", C.c_int32),
                ("rows_offset", C.c_int32), ("max_orb_distance_da", C.c_float)]


class CipParams(C.Structure):
    """ssk_cip_params: change-in-pose filter ([LEAST_SQUARES] and residual_th of the reference ini)."""
    _fields_ = [(n, C.c_double) for n in ("fx_l", "fy_l", "cx_l", "cy_l", "fx_r", "fy_r", "cx_r", "cy_r", "baseline")] + [
        ("initial_max_iters", C.c_int32), ("max_iters", C.c_int32), ("max_incr_cost", C.c_int32), ("use_robust_kernel", C.c_int32),
        ("residual_threshold", C.c_double), ("min_mod_out_vector", C.c_double), ("kernel_param", C.c_double),
        ("bad_tracking_th", C.c_int32), ("reserved", C.c_int32), ("residual_th", C.c_double)]


class CipResult(C.Structure):
    _fields_ = [("pose", C.c_double * 12), ("cost", C.c_double), ("valid", C.c_int32), ("iters_initial", C.c_int32),
                ("iters_final", C.c_int32), ("n_pairs", C.c_int32), ("n_inliers", C.c_int32), ("n_rejected", C.c_int32)]


class FmParams(C.Structure):
    """ssk_fm_params: fundamental-matrix RANSAC filter."""
    _fields_ = [("threshold", C.c_double), ("confidence", C.c_double), ("norm_cx", C.c_double), ("norm_cy", C.c_double),
                ("norm_scale", C.c_double), ("seed", C.c_uint64), ("max_iters", C.c_int32), ("min_points", C.c_int32)]


class FmResult(C.Structure):
    _fields_ = [("F", C.c_double * 9), ("valid", C.c_int32), ("n_pairs", C.c_int32), ("n_inliers", C.c_int32),
                ("n_rejected", C.c_int32), ("hypotheses", C.c_int32), ("best_hypothesis", C.c_int32)]


class DaStagedOut(C.Structure):
    """ssk_da_staged_out: tracked counts after each stage of ssk_data_association_staged, and the two filters' results."""
    _fields_ = [("tracked_da", C.c_int32), ("tracked_fm", C.c_int32), ("tracked_final", C.c_int32), ("reserved", C.c_int32),
                ("fm", FmResult), ("cip", CipResult)]


class FrontendParams(C.Structure):
    _fields_ = [("n_feats", C.c_int32), ("fast_threshold", C.c_int32), ("n_levels", C.c_int32),
                ("nms_radius", C.c_int32), ("orb_max_distance", C.c_int32), ("max_y_diff", C.c_float),
                ("min_disparity", C.c_float), ("max_disparity", C.c_float), ("harris_k", C.c_float)]


class KfOut(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("kp_left", C.c_void_p), ("kp_right", C.c_void_p),
                ("desc_left", C.c_void_p), ("desc_right", C.c_void_p), ("matches", C.c_void_p),
                ("n_left", C.c_int32), ("n_right", C.c_int32), ("n_matches", C.c_int32)]


def alloc_kf_out(cap):
    arrs = dict(kp_left=np.zeros(cap, KEYPOINT_DTYPE), kp_right=np.zeros(cap, KEYPOINT_DTYPE),
                desc_left=np.zeros((cap, 32), np.uint8), desc_right=np.zeros((cap, 32), np.uint8),
                matches=np.zeros(cap, DMATCH_DTYPE))
    o = KfOut(cap, arrs["kp_left"].ctypes.data, arrs["kp_right"].ctypes.data, arrs["desc_left"].ctypes.data,
              arrs["desc_right"].ctypes.data, arrs["matches"].ctypes.data, 0, 0, 0)
    return o, arrs


def trim_kf_out(o, arrs):
    return dict(kp_left=arrs["kp_left"][:o.n_left].copy(), kp_right=arrs["kp_right"][:o.n_right].copy(),
                desc_left=arrs["desc_left"][:o.n_left].copy(), desc_right=arrs["desc_right"][:o.n_right].copy(),
                matches=arrs["matches"][:o.n_matches].copy())


class KfBatch:
    """ssk_kfbatch wrapper: B keyframes resident in HBM (images, keypoints, descriptors, matches)."""

    def __init__(self, ctx, n_kf, width, height, max_feats=512):
        self.ctx, self.n_kf, self.w, self.h, self.max_feats = ctx, n_kf, width, height, max_feats
        h = C.c_void_p()
        ctx._chk(ctx.lib.ssk_kfbatch_create(ctx.h, n_kf, width, height, max_feats, C.byref(h)))
        self.h_ = h
        img = C.c_void_p()
        pitch = C.c_int()
        ctx._chk(ctx.lib.ssk_kfbatch_images(self.h_, C.byref(img), C.byref(pitch)))
        self.d_images, self.pitch = img.value, pitch.value
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h_", None):
            self.ctx.lib.ssk_kfbatch_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def image_tensor(self):
        """torch uint8 view [n_kf, 2, H, pitch] of the device image storage (no copy)."""
        import torch

        class _Arr:
            pass
        a = _Arr()
        a.__cuda_array_interface__ = {"shape": (self.n_kf, 2, self.h, self.pitch), "typestr": "|u1",
                                      "data": (self.d_images, False), "version": 3}
        return torch.as_tensor(a, device="cuda")

    def upload(self, kf, left, right):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        assert left.shape == (self.h, self.w) and right.shape == (self.h, self.w)
        self.ctx._chk(self.ctx.lib.ssk_kfbatch_upload(self.ctx.h, self.h_, int(kf), _ptr(left), _ptr(right), self.w))

    def run_frontend(self, prm, kf_begin=0, kf_end=None):
        self.ctx._chk(self.ctx.lib.ssk_frontend_run_batch(self.ctx.h, self.h_, int(kf_begin),
                                                          int(self.n_kf if kf_end is None else kf_end), C.byref(prm)))

    def download(self, kf):
        o, arrs = alloc_kf_out(self.max_feats)
        self.ctx._chk(self.ctx.lib.ssk_kfbatch_download(self.ctx.h, self.h_, int(kf), C.byref(o)))
        return trim_kf_out(o, arrs)

    def counts(self):
        nkp = np.zeros(2 * self.n_kf, np.int32)
        nm = np.zeros(self.n_kf, np.int32)
        self.ctx._chk(self.ctx.lib.ssk_kfbatch_counts(self.ctx.h, self.h_, _ptr(nkp), _ptr(nm)))
        return nkp.reshape(-1, 2), nm

    # -- bag of words (ssk_bow.hip) --
    def build_observations(self, lm_params, d_obs_px=None, d_init_pos=None, kf_begin=0, kf_end=None):
        """a10 / a11 for the whole batch, device to device: d_obs_px [n_kf, max_feats, 4] f32, d_init_pos [n_kf, max_feats, 3] f64 (torch tensors)."""
        e = self.n_kf if kf_end is None else kf_end
        self.ctx._chk(self.ctx.lib.ssk_build_observations_batch(self.ctx.h, self.h_, int(kf_begin), int(e), C.byref(lm_params),
                                                                None if d_obs_px is None else _dev(d_obs_px), None if d_init_pos is None else _dev(d_init_pos)))

    def bow_transform(self, voc, kf_begin=0, kf_end=None, ctx=None):
        """ctx: enqueue on another context's stream (same device) instead of the batch's own."""
        c = ctx or self.ctx
        c._chk(c.lib.ssk_bow_transform_batch(c.h, voc.h_, self.h_, int(kf_begin), int(self.n_kf if kf_end is None else kf_end)))

    def bow_download(self, kf):
        w = np.zeros(self.max_feats, np.int32)
        v = np.zeros(self.max_feats)
        n = C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_kfbatch_bow_download(self.ctx.h, self.h_, int(kf), _ptr(w), _ptr(v), self.max_feats, C.byref(n)))
        return w[:n.value].copy(), v[:n.value].copy()

    def bow_query(self, db, max_results=4, max_ids=None, kf_begin=0, kf_end=None, ctx=None):
        kf_end = self.n_kf if kf_end is None else kf_end
        mi = None if max_ids is None else np.ascontiguousarray(max_ids, np.int32)
        assert mi is None or len(mi) == kf_end - kf_begin
        c = ctx or self.ctx
        c._chk(c.lib.ssk_bow_db_query_batch(c.h, db.h_, self.h_, int(kf_begin), int(kf_end), int(max_results),
                                            None if mi is None else _ptr(mi)))

    def bow_results(self, kf_begin=0, kf_end=None):
        """[(ids, scores)] of the last bow_query for keyframes [kf_begin, kf_end)."""
        kf_end = self.n_kf if kf_end is None else kf_end
        nq = kf_end - kf_begin
        ids = np.zeros((nq, BOW_MAX_RESULTS), np.int32)
        sc = np.zeros((nq, BOW_MAX_RESULTS))
        n = np.zeros(nq, np.int32)
        self.ctx._chk(self.ctx.lib.ssk_kfbatch_bow_results(self.ctx.h, self.h_, int(kf_begin), int(kf_end), _ptr(ids), _ptr(sc), _ptr(n)))
        return [(ids[i, :n[i]].copy(), sc[i, :n[i]].copy()) for i in range(nq)]

    def debug_candidates(self, kf, eye):
        cap = (self.w // 2 + 1) * (self.h // 2 + 1)
        keys = np.zeros(cap, np.uint32)
        n = C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_kfbatch_debug_candidates(self.ctx.h, self.h_, int(kf), int(eye), _ptr(keys), cap, C.byref(n)))
        return keys[:n.value].copy()


BOW_MAX_RESULTS = 16


class BowVocab:
    """ssk_bow_vocab: the vocabulary tree on the device.  v: dict(k, L, weighting, scoring, node_id, parent_id, weight,
    desc[n,32], word_node) in file order (what host/CBoWManager.h parses from the DBoW2 YAML)."""

    def __init__(self, ctx, v):
        self.ctx = ctx
        nid = np.ascontiguousarray(v["node_id"], np.int32); pid = np.ascontiguousarray(v["parent_id"], np.int32)
        w = np.ascontiguousarray(v["weight"], np.float64); d = np.ascontiguousarray(v["desc"], np.uint8).reshape(-1, 32)
        wn = np.ascontiguousarray(v["word_node"], np.int32)
        h = C.c_void_p()
        ctx._chk(ctx.lib.ssk_bow_vocab_create(ctx.h, int(v["k"]), int(v["L"]), int(v.get("weighting", 0)), int(v.get("scoring", 0)),
                                              len(nid), _ptr(nid), _ptr(pid), _ptr(w), _ptr(d), len(wn), _ptr(wn), C.byref(h)))
        self.h_ = h
        self.n_words = len(wn)
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h_", None):
            self.ctx.lib.ssk_bow_vocab_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        k, L, nn, nw = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_bow_vocab_info(self.h_, C.byref(k), C.byref(L), C.byref(nn), C.byref(nw)))
        return k.value, L.value, nn.value, nw.value

    def transform(self, desc):
        """Host drop-in: descriptors [n, 32] -> (ascending word ids, L1-normalised values)."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        w = np.zeros(len(d) + 1, np.int32)
        v = np.zeros(len(d) + 1)
        n = C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_bow_transform(self.ctx.h, self.h_, _ptr(d), len(d), _ptr(w), _ptr(v), C.byref(n)))
        return w[:n.value].copy(), v[:n.value].copy()


class BowDb:
    """ssk_bow_db: bag-of-words database in HBM (entry ids = insertion order)."""

    def __init__(self, ctx, voc, capacity_entries, max_words_per_entry=512):
        self.ctx, self.voc = ctx, voc
        h = C.c_void_p()
        ctx._chk(ctx.lib.ssk_bow_db_create(ctx.h, voc.h_, int(capacity_entries), int(max_words_per_entry), C.byref(h)))
        self.h_ = h
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h_", None):
            self.ctx.lib.ssk_bow_db_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.ctx.lib.ssk_bow_db_size(self.h_)

    def clear(self):
        self.ctx._chk(self.ctx.lib.ssk_bow_db_clear(self.ctx.h, self.h_))

    def add(self, words, vals):
        w = np.ascontiguousarray(words, np.int32); v = np.ascontiguousarray(vals, np.float64)
        e = C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_bow_db_add(self.ctx.h, self.h_, _ptr(w), _ptr(v), len(w), C.byref(e)))
        return e.value

    def add_from_batch(self, batch, kf):
        e = C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_bow_db_add_from_batch(self.ctx.h, self.h_, batch.h_, int(kf), C.byref(e)))
        return e.value

    def query(self, words, vals, max_results=4, max_id=-1):
        w = np.ascontiguousarray(words, np.int32); v = np.ascontiguousarray(vals, np.float64)
        ids = np.zeros(BOW_MAX_RESULTS, np.int32); sc = np.zeros(BOW_MAX_RESULTS)
        n = C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_bow_db_query(self.ctx.h, self.h_, _ptr(w), _ptr(v), len(w), int(max_results), int(max_id),
                                                    _ptr(ids), _ptr(sc), C.byref(n)))
        return ids[:n.value].copy(), sc[:n.value].copy()


class PgParams(C.Structure):
    """ssk_pg_params; defaults: MRPT's 100 iterations / tau 1e-3, CG to 1e-10; solver 0 = direct (spanning tree + Woodbury)
    when it applies, 1 = CG always, 2 = direct or an error."""
    _fields_ = [("max_iters", C.c_int32), ("pcg_max_iters", C.c_int32), ("tau", C.c_double), ("max_lambda", C.c_double),
                ("converged_rel_decrease", C.c_double), ("pcg_tol", C.c_double), ("solver", C.c_int32), ("reserved", C.c_int32)]

    def __init__(self, max_iters=100, pcg_max_iters=400, tau=1e-3, max_lambda=1e20, converged_rel_decrease=1e-9, pcg_tol=1e-10, solver=0):
        super().__init__(max_iters, pcg_max_iters, tau, max_lambda, converged_rel_decrease, pcg_tol, solver, 0)


class PgResult(C.Structure):
    _fields_ = [("iters", C.c_int32), ("accepted", C.c_int32), ("stop_reason", C.c_int32), ("pcg_iters_total", C.c_int32),
                ("cost_init", C.c_double), ("cost_final", C.c_double), ("lambda_final", C.c_double)]


class RbaOptions(C.Structure):
    """ssk_rba_options (include/ssk_rba.h); defaults = demo/config_imgdir_kitti_srba.ini."""
    _fields_ = [(n, C.c_double) for n in ("fx_l", "fy_l", "cx_l", "cy_l", "fx_r", "fy_r", "cx_r", "cy_r", "baseline")] + [
        ("sensor_pose", C.c_double * 12), ("max_tree_depth", C.c_int32), ("max_optimize_depth", C.c_int32), ("submap_size", C.c_int32),
        ("min_obs_to_loop_closure", C.c_int32), ("use_robust_kernel", C.c_int32), ("use_robust_kernel_stage1", C.c_int32),
        ("reserved", C.c_int32), ("kernel_param", C.c_double)]

    @classmethod
    def kitti(cls, **over):
        o = cls(707.0912, 707.0912, 601.8873, 183.1104, 707.0912, 707.0912, 601.8873, 183.1104, 0.54)
        o.sensor_pose = (C.c_double * 12)(0, 0, 1, 0, -1, 0, 0, 0, 0, -1, 0, 0)        # camera z forward, x right, y down
        o.max_tree_depth, o.max_optimize_depth, o.submap_size, o.min_obs_to_loop_closure = 4, 5, 10, 6
        o.use_robust_kernel, o.use_robust_kernel_stage1, o.kernel_param = 1, 1, 1.5
        for k, v in over.items():
            setattr(o, k, v)
        return o


RBA_OBS_DTYPE = np.dtype([("feat_id", "<u8"), ("left_x", "<f4"), ("left_y", "<f4"), ("right_x", "<f4"), ("right_y", "<f4"), ("rel_pos", "<f8", 3)])


class RbaKfInfo(C.Structure):
    _fields_ = [("kf_id", C.c_int32), ("n_created_edges", C.c_int32), ("created_edge_ids", C.c_int32 * 16),
                ("total_sqr_error_init", C.c_double), ("total_sqr_error_final", C.c_double), ("stg1_obs_rmse", C.c_double),
                ("num_observations", C.c_int32), ("num_iters", C.c_int32)]


class RbaBatchTiming(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("bookkeeping_and_windows_ms", "stage1_create_ms", "stage1_solve_ms", "stage2_refresh_ms", "stage2_create_ms",
                                          "stage2_solve_ms", "writeback_ms", "stage1_symbolic_ms", "stage2_symbolic_ms", "stage1_mean_obs", "stage2_mean_obs",
                                          "stage2_mean_edges", "stage1_mean_iters", "stage2_mean_iters")] + \
               [(k, C.c_int32) for k in ("host_threads", "n_stage1", "n_stage2", "stage2_max_edges", "stage2_max_obs", "reserved")] + \
               [("stage1_kernel", C.c_char * 96), ("stage2_kernel", C.c_char * 96), ("teardown_ms", C.c_double), ("total_ms", C.c_double)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        d["stage1_kernel"] = self.stage1_kernel.decode(); d["stage2_kernel"] = self.stage2_kernel.decode()
        return d


def rba_define_kf_batch(engines, obs_lists, optimize=True, handles=None):
    """ssk_rba_define_kf_batch: the engines (one context) advanced in lock-step, engine i with obs_lists[i].  Returns (infos, timing dict)."""
    n = len(engines)
    lib = engines[0].lib
    if isinstance(obs_lists, tuple):                           # (records [n, cap] of RBA_OBS_DTYPE, counts [n]): rows used as they are, no copies
        arr2, ns = obs_lists
        assert arr2.dtype == RBA_OBS_DTYPE and arr2.flags.c_contiguous and arr2.shape[0] == n
        ns = np.ascontiguousarray(ns, np.int32)
        ps = (C.c_void_p * n)(*[arr2.ctypes.data + i * arr2.strides[0] for i in range(n)])
    else:
        arrs = [np.ascontiguousarray(o, RBA_OBS_DTYPE) for o in obs_lists]
        ps = (C.c_void_p * n)(*[a.ctypes.data if len(a) else None for a in arrs])
        ns = np.array([len(a) for a in arrs], np.int32)
    hs = handles if handles is not None else (engines if isinstance(engines, C.Array) else (C.c_void_p * n)(*[e.h_ for e in engines]))
    infos = (RbaKfInfo * n)()
    tm = RbaBatchTiming()
    engines[0].ctx.lib.ssk_env_refresh(engines[0].ctx.h)
    engines[0]._chk(lib.ssk_rba_define_kf_batch(hs, n, ps, _ptr(ns), int(optimize), infos, C.byref(tm)))
    return list(infos), tm.as_dict()


_FLIB = None


def load_facade_library():
    """libssk_facade.so (C++ facade + the C-ABI of include/ssk_rba.h), on top of libssk_hip.so."""
    global _FLIB
    if _FLIB is None:
        load_library()
        p = os.path.join(HERE, "libssk_facade.so")
        if not os.path.exists(p):
            raise ImportError("libssk_facade.so is not built: run `python __graft_entry__.py build`")
        _FLIB = C.CDLL(p)
        _FLIB.ssk_rba_last_error.restype = C.c_char_p
    return _FLIB


class Rba:
    """ssk_rba: the relative-BA engine behind rba.define_new_keyframe (include/ssk_rba.h)."""

    def __init__(self, ctx, options):
        self.ctx, self.lib = ctx, load_facade_library()
        h = C.c_void_p()
        rc = self.lib.ssk_rba_create(ctx.h, C.byref(options), C.byref(h))
        if rc != 0:
            raise SskError(rc, "ssk_rba_create failed")
        self.h_ = h
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h_", None):
            self.lib.ssk_rba_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SskError(rc, self.lib.ssk_rba_last_error(self.h_).decode() or self.ctx.lib.ssk_last_error(self.ctx.h).decode())

    def set_initial_kf_pose(self, pose):
        p = np.ascontiguousarray(pose, np.float64).reshape(12)
        self._chk(self.lib.ssk_rba_set_initial_kf_pose(self.h_, _ptr(p)))

    def define_kf(self, obs, optimize=True):
        o = np.ascontiguousarray(obs, RBA_OBS_DTYPE)
        info = RbaKfInfo()
        self.ctx.lib.ssk_env_refresh(self.ctx.h)               # (tests switch SSK_LM_* on a live context)
        self._chk(self.lib.ssk_rba_define_kf(self.h_, _ptr(o), len(o), int(optimize), C.byref(info)))
        return info

    def counts(self):
        return self.lib.ssk_rba_num_keyframes(self.h_), self.lib.ssk_rba_num_edges(self.h_), self.lib.ssk_rba_num_landmarks(self.h_)

    def localmap_centers(self, kf):
        out = np.zeros(16, np.int32); n = C.c_int32()
        self._chk(self.lib.ssk_rba_localmap_centers(self.h_, int(kf), _ptr(out), 16, C.byref(n)))
        return [int(v) for v in out[:n.value]]

    def localmap_kfs(self, center):
        """keyframes of the local map centred at `center`, or None when it is not a centre."""
        out = np.zeros(256, np.int32); n = C.c_int32()
        self._chk(self.lib.ssk_rba_localmap_kfs(self.h_, int(center), _ptr(out), 256, C.byref(n)))
        return None if n.value < 0 else [int(v) for v in out[:n.value]]

    def edge(self, e):
        f, t = C.c_int32(), C.c_int32()
        p = np.zeros(12)
        self._chk(self.lib.ssk_rba_get_edge(self.h_, int(e), C.byref(f), C.byref(t), _ptr(p)))
        return f.value, t.value, p

    def landmark(self, feat_id):
        b = C.c_int32()
        p = np.zeros(3)
        self.lib.ssk_rba_get_landmark.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        rc = self.lib.ssk_rba_get_landmark(self.h_, int(feat_id), C.byref(b), _ptr(p))
        return None if rc != 0 else (b.value, p)

    def spanning_tree_pose(self, root, target, max_depth=1000):
        p = np.zeros(12)
        found = C.c_int32()
        self._chk(self.lib.ssk_rba_spanning_tree_pose(self.h_, int(root), int(target), int(max_depth), _ptr(p), C.byref(found)))
        return p if found.value else None

    def global_poses(self, root=0):
        nk, ne, _ = self.counts()
        X = np.zeros((max(nk, 1), 12)); ij = np.zeros((max(ne, 1), 2), np.int32); z = np.zeros((max(ne, 1), 12))
        nn, nm = C.c_int32(), C.c_int32()
        self._chk(self.lib.ssk_rba_global_poses(self.h_, int(root), _ptr(X), len(X), _ptr(ij), _ptr(z), len(z), C.byref(nn), C.byref(nm)))
        return X[:nn.value].copy(), ij[:nm.value].copy(), z[:nm.value].copy()


class SlamFrameReport(C.Structure):
    _fields_ = [("frame", C.c_int32), ("vo_valid", C.c_int32), ("tracked_from_last_frame", C.c_int32), ("tracked_from_last_kf", C.c_int32),
                ("vo_pose", C.c_double * 12), ("checked_for_kf", C.c_int32), ("inserted_kf", C.c_int32), ("vo_force_new_kf", C.c_int32),
                ("vo_force_check_tracking", C.c_int32), ("vo_force_check_distance", C.c_int32), ("potential_loop_closure", C.c_int32),
                ("confirmed_loop_closure", C.c_int32), ("lc_kf_id", C.c_int32), ("n_query", C.c_int32), ("query_ids", C.c_int32 * 4),
                ("query_scores", C.c_double * 4), ("n_similar", C.c_int32), ("similar_kfs", C.c_int32 * 8), ("highest_tracked", C.c_int32),
                ("kf_id", C.c_int32), ("n_created_edges", C.c_int32), ("total_sqr_error_init", C.c_double), ("total_sqr_error_final", C.c_double),
                ("n_keyframes", C.c_int32), ("reserved", C.c_int32), ("lm_stage2_edges", C.c_int32), ("lm_stage2_obs", C.c_int32)]


class SlamTiming(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("vo_ms", "query_ms", "da_ms", "insert_ms", "lm_window_build_ms", "lm_batch_create_ms", "lm_solve_ms", "lm_readback_ms")] + \
               [(n, C.c_int32) for n in ("frames", "kf_frames", "lm_solves", "lm_iters", "lm_max_opt_edges", "lm_max_obs")]


class Slam:
    """ssk_slam: the estimator (include/ssk_slam.h): one call per stereo frame, the reference's whole loop behind it."""

    def __init__(self, ctx, width, height, preset="kitti", **options):
        self.ctx, self.lib = ctx, load_facade_library()
        self.lib.ssk_slam_last_error.restype = C.c_char_p
        self.lib.ssk_slam_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        h = C.c_void_p()
        rc = self.lib.ssk_slam_create(ctx.h, preset.encode(), int(width), int(height), C.byref(h))
        if rc != 0:
            raise SskError(rc, "ssk_slam_create failed")
        self.h_, self.w, self.hgt = h, width, height
        _LIVE.add(self)
        for k, v in options.items():
            self._chk(self.lib.ssk_slam_set_option(self.h_, k.encode(), float(v)))

    def close(self):
        if getattr(self, "h_", None):
            self.lib.ssk_slam_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SskError(rc, self.lib.ssk_slam_last_error(self.h_).decode() or self.ctx.lib.ssk_last_error(self.ctx.h).decode())

    def load_vocabulary(self, path):
        self._chk(self.lib.ssk_slam_load_vocabulary(self.h_, path.encode()))

    def process_frame(self, left, right):
        l = np.ascontiguousarray(left, np.uint8); r = np.ascontiguousarray(right, np.uint8)
        assert l.shape == (self.hgt, self.w) and r.shape == (self.hgt, self.w)
        rep = SlamFrameReport()
        self._chk(self.lib.ssk_slam_process_frame(self.h_, _ptr(l), _ptr(r), self.w, C.byref(rep)))
        return rep

    def timing(self):
        t = SlamTiming()
        self._chk(self.lib.ssk_slam_get_timing(self.h_, C.byref(t)))
        return t

    def set_debug_dir(self, path):
        self._chk(self.lib.ssk_slam_set_debug_dir(self.h_, str(path).encode()))

    def get_keyframe(self, kf, cap=1024):
        """(dict as Context.frontend_run returns it, match ids u64[n_matches]) of keyframe `kf`."""
        o, arrs = alloc_kf_out(cap)
        ids = np.zeros(cap, np.uint64)
        self._chk(self.lib.ssk_slam_get_keyframe(self.h_, int(kf), C.byref(o), _ptr(ids)))
        d = trim_kf_out(o, arrs)
        return d, ids[:o.n_matches].copy()

    def last_da(self, which, cap=1024):
        """t_kf_da_info of the last keyframe check against its `which`-th similar keyframe."""
        other, tracked, pos, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        first = np.zeros(cap, np.int32); dist = np.zeros(cap, np.float64); init = np.zeros(12)
        self._chk(self.lib.ssk_slam_last_da(self.h_, int(which), C.byref(other), C.byref(tracked), C.byref(pos), _ptr(first), _ptr(dist), cap,
                                            C.byref(n), _ptr(init)))
        return dict(other_kf=other.value, tracked=tracked.value, sorted_pos=pos.value, tracking_first=first[:n.value].copy(),
                    tracking_dist=dist[:n.value].copy(), init_pose=init)

    def counts(self):
        return self.lib.ssk_slam_num_keyframes(self.h_), self.lib.ssk_slam_num_edges(self.h_), self.lib.ssk_slam_num_landmarks(self.h_)

    def optimize_pose_graph(self, path=None):
        n = max(self.counts()[0], 1)
        X = np.zeros((n, 12))
        cnt = C.c_int32()
        info = PgResult()
        self._chk(self.lib.ssk_slam_optimize_pose_graph(self.h_, _ptr(X), n, C.byref(cnt), C.byref(info), path.encode() if path else None))
        return X[:cnt.value].copy(), info


class LmWindow(C.Structure):
    _fields_ = [("n_edges_total", C.c_int32), ("edge_unknown_slot", C.c_void_p), ("edge_inv_pose", C.c_void_p),
                ("n_lms_total", C.c_int32), ("lm_unknown_slot", C.c_void_p), ("lm_pos", C.c_void_p),
                ("n_obs", C.c_int32), ("obs_lm", C.c_void_p), ("obs_px", C.c_void_p),
                ("obs_path_off", C.c_void_p), ("path_edge", C.c_void_p)]


# ssk_lm_window as a numpy record (C layout): a batch's array of windows filled column by column instead of one ctypes struct at a time
LM_WINDOW_DTYPE = np.dtype([("n_edges_total", "<i4"), ("edge_unknown_slot", "<u8"), ("edge_inv_pose", "<u8"), ("n_lms_total", "<i4"), ("lm_unknown_slot", "<u8"),
                            ("lm_pos", "<u8"), ("n_obs", "<i4"), ("obs_lm", "<u8"), ("obs_px", "<u8"), ("obs_path_off", "<u8"), ("path_edge", "<u8")], align=True)
assert LM_WINDOW_DTYPE.itemsize == C.sizeof(LmWindow) and all(LM_WINDOW_DTYPE.fields[n][1] == getattr(LmWindow, n).offset for n, _ in LmWindow._fields_)


def lm_window_array(rec):
    """the ssk_lm_window array over a numpy record array of LM_WINDOW_DTYPE (shares its memory: keep `rec` alive)"""
    assert rec.dtype == LM_WINDOW_DTYPE and rec.flags.c_contiguous
    return (LmWindow * len(rec)).from_buffer(rec)


class LmParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("fx_l", "fy_l", "cx_l", "cy_l", "fx_r", "fy_r", "cx_r", "cy_r", "baseline")] + \
               [("sensor_pose", C.c_double * 12), ("max_iters", C.c_int32), ("use_robust_kernel", C.c_int32),
                ("kernel_param", C.c_double), ("tau", C.c_double), ("max_lambda", C.c_double),
                ("max_error_per_obs_to_stop", C.c_double), ("converged_rel_decrease", C.c_double)]


class LmResult(C.Structure):
    _fields_ = [("iters", C.c_int32), ("accepted", C.c_int32), ("stop_reason", C.c_int32), ("n_obs", C.c_int32),
                ("err_init", C.c_double), ("err_final", C.c_double), ("lambda_final", C.c_double)]


class LmTraceEntry(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("err_before", C.c_double), ("err_after", C.c_double),
                ("delta_p_norm", C.c_double), ("delta_f_norm", C.c_double), ("accepted", C.c_int32),
                ("cholesky_ok", C.c_int32)]


class LmBatch:
    """ssk_lm_batch wrapper.  `windows`: objects with the ssk_lm_window arrays as numpy attributes
    (edge_unknown_slot, edge_inv_pose, lm_unknown_slot, lm_pos, obs_lm, obs_px, obs_path_off, path_edge)."""

    @staticmethod
    def marshal(windows):
        """(ssk_lm_window array, the numpy arrays it points into): the C view of a list of windows, reusable for several batches"""
        windows = list(windows)
        arr = (LmWindow * len(windows))()
        keep = []
        for i, w in enumerate(windows):
            a = [np.ascontiguousarray(w.edge_unknown_slot, np.int32), np.ascontiguousarray(w.edge_inv_pose, np.float64),
                 np.ascontiguousarray(w.lm_unknown_slot, np.int32), np.ascontiguousarray(w.lm_pos, np.float64),
                 np.ascontiguousarray(w.obs_lm, np.int32), np.ascontiguousarray(w.obs_px, np.float32),
                 np.ascontiguousarray(w.obs_path_off, np.int32), np.ascontiguousarray(w.path_edge, np.int32)]
            keep.append(a)
            p = lambda x: x.ctypes.data if x.size else 0
            arr[i] = LmWindow(len(a[0]), p(a[0]), p(a[1]), len(a[2]), p(a[2]), p(a[3]), len(a[4]), p(a[4]), p(a[5]), p(a[6]), p(a[7]))
        return arr, keep

    def __init__(self, ctx, windows, trace=False, marshalled=None):
        self.ctx, self.windows = ctx, list(windows)
        n = len(self.windows)
        arr, self._keep = marshalled if marshalled is not None else LmBatch.marshal(self.windows)
        h = C.c_void_p()
        # the library reads its environment switches once per context (ssk_create); tests and A/B tools switch SSK_LM_KERNEL & co. on a live context
        ctx.lib.ssk_env_refresh(ctx.h)
        t0 = time.perf_counter()
        ctx._chk(ctx.lib.ssk_lm_batch_create(ctx.h, arr, n, int(bool(trace)), C.byref(h)))
        self.create_ms = 1e3 * (time.perf_counter() - t0)      # the C call alone: symbolic build + allocation + upload (no Python marshalling)
        self.h_, self.n = h, n
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h_", None):
            self.ctx.lib.ssk_lm_batch_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def reset(self, ctx=None):
        """`ctx`: the context (stream) to enqueue on; default: the one the batch was created with"""
        c = ctx or self.ctx
        c._chk(c.lib.ssk_lm_batch_reset(c.h, self.h_))

    def run(self, prm, ctx=None):
        c = ctx or self.ctx
        c.lib.ssk_env_refresh(c.h)                             # (test hooks of the cooperative kernel are switched between runs)
        c._chk(c.lib.ssk_lm_batch_run(c.h, self.h_, C.byref(prm)))

    def kernel_name(self):
        buf = C.create_string_buffer(192)
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_kernel_name(self.ctx.h, self.h_, buf, 192))
        return buf.value.decode()

    def solver_name(self):
        buf = C.create_string_buffer(96)
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_solver_name(self.ctx.h, self.h_, buf, 96))
        return buf.value.decode()

    def symbolic_ms(self):
        """(host milliseconds of the symbolic build in ssk_lm_batch_create, host threads used)"""
        ms, thr = C.c_double(0), C.c_int(0)
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_symbolic_ms(self.ctx.h, self.h_, C.byref(ms), C.byref(thr)))
        return ms.value, thr.value

    def set_export(self, on=True):
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_set_export(self.ctx.h, self.h_, int(bool(on))))

    def pack_shared(self, w, d_lm_idx, prm, d_out):
        """window w's blocks (position 3, g 3, upper H 6) of the shared landmarks d_lm_idx (device int32: landmark index in the window or -1)
        at the window's current state -> d_out (device f64 [n_shared * 12])."""
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_pack_shared(self.ctx.h, self.h_, int(w), int(d_lm_idx.numel()), _dev(d_lm_idx), C.byref(prm), _dev(d_out)))

    def set_shared(self, w, d_lm_idx, d_pos):
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_set_shared(self.ctx.h, self.h_, int(w), int(d_lm_idx.numel()), _dev(d_lm_idx), _dev(d_pos)))

    def results(self, check=True):
        """check=False: the per-window records as the call filled them even when it reports an error (stop_reason 5)."""
        out = (LmResult * self.n)()
        rc = self.ctx.lib.ssk_lm_batch_results(self.ctx.h, self.h_, out)
        if check:
            self.ctx._chk(rc)
        return list(out)

    def state(self, w):
        win = self.windows[w]
        poses = np.zeros((len(win.edge_unknown_slot), 12))
        lms = np.zeros((len(win.lm_unknown_slot), 3))
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_get_state(self.ctx.h, self.h_, int(w), _ptr(poses), _ptr(lms)))
        return poses, lms

    def set_state(self, w, poses=None, lms=None):
        """ssk_lm_batch_set_state: replaces the edge poses [n_edges, 12] and / or landmark positions [n_lms, 3] of window w (None: kept)."""
        p = None if poses is None else np.ascontiguousarray(poses, dtype=np.float64)
        l = None if lms is None else np.ascontiguousarray(lms, dtype=np.float64)
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_set_state(self.ctx.h, self.h_, int(w), _ptr(p) if p is not None else None, _ptr(l) if l is not None else None))

    def trace(self, w, cap=64):
        win = self.windows[w]
        E = int((np.asarray(win.edge_unknown_slot) >= 0).sum())
        L = int((np.asarray(win.lm_unknown_slot) >= 0).sum())
        ent = (LmTraceEntry * cap)()
        dp = np.zeros(cap * 6 * E + 1)
        df = np.zeros(cap * 3 * L + 1)
        n = C.c_int()
        self.ctx._chk(self.ctx.lib.ssk_lm_batch_get_trace(self.ctx.h, self.h_, int(w), ent, cap, _ptr(dp), _ptr(df), C.byref(n)))
        k = n.value
        return list(ent)[:k], dp[:k * 6 * E].reshape(k, 6 * E), df[:k * 3 * L].reshape(k, 3 * L)


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


def _dev(p):
    """device pointer argument: int address, torch tensor (data_ptr) or None."""
    if p is None:
        return C.c_void_p(0)
    if hasattr(p, "data_ptr"):
        return C.c_void_p(p.data_ptr())
    return C.c_void_p(int(p))


def make_kf_view(kp_left, kp_right, desc_left, desc_right, matches):
    """Returns (view, keepalive): keep the tuple alive while the view is in use."""
    kp_left = np.ascontiguousarray(kp_left, dtype=KEYPOINT_DTYPE)
    kp_right = np.ascontiguousarray(kp_right, dtype=KEYPOINT_DTYPE)
    desc_left = np.ascontiguousarray(desc_left, dtype=np.uint8).reshape(-1, 32)
    desc_right = np.ascontiguousarray(desc_right, dtype=np.uint8).reshape(-1, 32)
    matches = np.ascontiguousarray(matches, dtype=DMATCH_DTYPE)
    v = KfView(_ptr(kp_left), len(kp_left), _ptr(kp_right), len(kp_right), _ptr(desc_left), _ptr(desc_right),
               _ptr(matches), len(matches))
    return v, (kp_left, kp_right, desc_left, desc_right, matches)


class Context:
    """ssk_ctx wrapper.  stream: a hipStream_t handle (int) or None for a context-owned stream."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.ssk_create(C.byref(h), int(device), C.c_void_p(stream or 0))
        if rc != 0:
            raise SskError(rc, self.lib.ssk_last_error(None).decode())
        self.h = h
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.lib.ssk_destroy(self.h)
            self.h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise SskError(rc, self.lib.ssk_last_error(self.h).decode())

    # -- context ---------------------------------------------------------------------------------
    def synchronize(self):
        self._chk(self.lib.ssk_synchronize(self.h))

    def stream(self):
        return self.lib.ssk_stream(self.h)

    def timer_start(self):
        self._chk(self.lib.ssk_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._chk(self.lib.ssk_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def direction_thresholds(self):
        out = np.zeros(19, np.float32)
        self._chk(self.lib.ssk_debug_direction_thresholds(self.h, _ptr(out)))
        return out

    # -- B2 ------------------------------------------------------------------------------------------
    def hamming_match(self, query, train):
        q = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        out = np.zeros(len(q), DMATCH_DTYPE)
        n = C.c_int()
        self._chk(self.lib.ssk_hamming_match(self.h, _ptr(q), len(q), _ptr(t), len(t), _ptr(out), C.byref(n)))
        return out[:n.value]

    def hamming_match_batched_dev(self, d_q, d_q_off, d_q_gather, d_t, d_t_off, d_t_gather, n_pairs, max_q,
                                  d_idx, d_dist):
        self._chk(self.lib.ssk_hamming_match_batched_dev(
            self.h, _dev(d_q), _dev(d_q_off), _dev(d_q_gather), _dev(d_t), _dev(d_t_off), _dev(d_t_gather),
            int(n_pairs), int(max_q), _dev(d_idx), _dev(d_dist)))

    def da_filters(self, matches, n_other, this_xy, other_xy, params):
        m = np.ascontiguousarray(matches, DMATCH_DTYPE)
        a = np.ascontiguousarray(this_xy, np.float32)
        b = np.ascontiguousarray(other_xy, np.float32)
        status = np.zeros(len(m), np.int32)
        tracked = C.c_int32()
        self._chk(self.lib.ssk_da_filters(self.h, _ptr(m), len(m), int(n_other), _ptr(a), _ptr(b),
                                          C.byref(params), _ptr(status), C.byref(tracked)))
        return status, tracked.value

    # -- B1 ------------------------------------------------------------------------------------------
    def set_brief_pattern(self, pattern):
        p = np.ascontiguousarray(pattern, np.int8).reshape(1024)
        self._chk(self.lib.ssk_set_brief_pattern(self.h, _ptr(p)))

    def set_orb_steering(self, mode):
        """C7: 1 (default) = cv::ORB's cos / sin of the fastAtan2 angle (include/ssk_sincos.h), 0 = exact centroid direction."""
        self._chk(self.lib.ssk_set_orb_steering(self.h, int(mode)))

    def get_brief_pattern(self):
        p = np.zeros(1024, np.int8)
        self._chk(self.lib.ssk_get_brief_pattern(self.h, _ptr(p)))
        return p

    def frontend_run(self, left, right, prm):
        left = np.ascontiguousarray(left, np.uint8)
        right = np.ascontiguousarray(right, np.uint8)
        h, w = left.shape
        o, arrs = alloc_kf_out(max(prm.n_feats, 1))
        self._chk(self.lib.ssk_frontend_run(self.h, _ptr(left), _ptr(right), w, h, w, C.byref(prm), C.byref(o)))
        return trim_kf_out(o, arrs)

    def da_run_batch(self, this_b, other_b, pairs, params, d_idx, d_dist, d_status, d_tracked):
        pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self._chk(self.lib.ssk_da_run_batch(self.h, this_b.h_, other_b.h_, _ptr(pr), len(pr), C.byref(params),
                                            _dev(d_idx), _dev(d_dist), _dev(d_status), _dev(d_tracked)))

    def cip_run_batch(self, this_b, other_b, pairs, params, d_idx, d_status, d_tracked, init_poses=None, d_results=None):
        pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        ip = None if init_poses is None else np.ascontiguousarray(init_poses, np.float64).reshape(len(pr), 12)
        self._chk(self.lib.ssk_cip_run_batch(self.h, this_b.h_, other_b.h_, _ptr(pr), len(pr), C.byref(params),
                                             None if ip is None else _ptr(ip), _dev(d_idx), _dev(d_status), _dev(d_tracked),
                                             None if d_results is None else _dev(d_results)))

    def fm_run_batch(self, this_b, other_b, pairs, params, d_idx, d_status, d_tracked, d_results=None):
        pr = np.ascontiguousarray(pairs, np.int32).reshape(-1, 2)
        self._chk(self.lib.ssk_fm_run_batch(self.h, this_b.h_, other_b.h_, _ptr(pr), len(pr), C.byref(params), _dev(d_idx), _dev(d_status),
                                            _dev(d_tracked), None if d_results is None else _dev(d_results)))

    def fm_filter(self, this_xy, other_xy, train_idx, status, params):
        """ssk_fm_filter on host arrays: returns (status after the filter, FmResult)."""
        tx = np.ascontiguousarray(this_xy, np.float32).reshape(-1, 2)
        ox = np.ascontiguousarray(other_xy, np.float32).reshape(-1, 2)
        ti = np.ascontiguousarray(train_idx, np.int32)
        st = np.ascontiguousarray(status, np.int32).copy()
        out = FmResult()
        self._chk(self.lib.ssk_fm_filter(self.h, _ptr(tx), len(tx), _ptr(ox), len(ox), _ptr(ti), _ptr(st), C.byref(params), C.byref(out)))
        return st, out

    def data_association_staged(self, this_view, other_view, params, fm_params=None, cip_params=None, init_pose=None):
        """ssk_data_association_staged: association + (optional) fundamental-matrix + (optional) change-in-pose filters of one pair
        in one call.  Returns (matches, status after the last stage, DaStagedOut)."""
        n = this_view.n_matches
        m = np.zeros(n, DMATCH_DTYPE)
        st = np.zeros(n, np.int32)
        out = DaStagedOut()
        T = None if init_pose is None else np.ascontiguousarray(init_pose, np.float64).reshape(12)
        self.lib.ssk_data_association_staged.argtypes = [C.c_void_p] * 10
        self._chk(self.lib.ssk_data_association_staged(self.h, C.byref(this_view), C.byref(other_view), C.byref(params),
                                                       None if fm_params is None else C.byref(fm_params),
                                                       None if cip_params is None else C.byref(cip_params),
                                                       None if T is None else _ptr(T), _ptr(m), _ptr(st), C.byref(out)))
        return m, st, out

    def data_association_staged_multi(self, this_view, other_views, params, fm_params=None, fm_seeds=None, cip_params=None, init_poses=None):
        """ssk_data_association_staged_multi: this keyframe against every keyframe of other_views in one call.
        Returns (matches [S, n], statuses [S, n], list of DaStagedOut)."""
        S, n = len(other_views), this_view.n_matches
        views = (KfView * S)(*other_views)
        m = np.zeros((S, n), DMATCH_DTYPE)
        st = np.zeros((S, n), np.int32)
        outs = (DaStagedOut * S)()
        seeds = None if fm_seeds is None else np.ascontiguousarray(fm_seeds, np.uint64)
        T = None if init_poses is None else np.ascontiguousarray(init_poses, np.float64).reshape(S, 12)
        self.lib.ssk_data_association_staged_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 8
        self._chk(self.lib.ssk_data_association_staged_multi(self.h, C.byref(this_view), views, S, C.byref(params),
                                                             None if fm_params is None else C.byref(fm_params), None if seeds is None else _ptr(seeds),
                                                             None if cip_params is None else C.byref(cip_params), None if T is None else _ptr(T),
                                                             _ptr(m), _ptr(st), outs))
        return m, st, list(outs)

    def da_fundamental_filter(self, this_view, other_view, da_matches, status, params):
        m = np.ascontiguousarray(da_matches, DMATCH_DTYPE)
        st = np.ascontiguousarray(status, np.int32).copy()
        out = FmResult()
        self._chk(self.lib.ssk_da_fundamental_filter(self.h, C.byref(this_view), C.byref(other_view), _ptr(m), _ptr(st), C.byref(params), C.byref(out)))
        return st, out

    def cip_filter(self, this_obs, other_obs, train_idx, status, init_pose, params):
        """ssk_cip_filter on host arrays: returns (status after the filter, residuals of the tracked pairs, CipResult)."""
        to = np.ascontiguousarray(this_obs, np.float32).reshape(-1, 4)
        oo = np.ascontiguousarray(other_obs, np.float32).reshape(-1, 4)
        ti = np.ascontiguousarray(train_idx, np.int32)
        st = np.ascontiguousarray(status, np.int32).copy()
        n_trk = int((st == 0).sum())
        T = np.ascontiguousarray(init_pose, np.float64).reshape(12)
        res = np.zeros(len(to) + 1)
        out = CipResult()
        self._chk(self.lib.ssk_cip_filter(self.h, _ptr(to), len(to), _ptr(oo), len(oo), _ptr(ti), _ptr(st), _ptr(T), C.byref(params),
                                          _ptr(res), C.byref(out)))
        return st, res[:n_trk].copy(), out

    def da_change_in_pose(self, this_view, other_view, da_matches, status, init_pose, params):
        m = np.ascontiguousarray(da_matches, DMATCH_DTYPE)
        st = np.ascontiguousarray(status, np.int32).copy()
        n_trk = int((st == 0).sum())
        T = np.ascontiguousarray(init_pose, np.float64).reshape(12)
        res = np.zeros(len(m) + 1)
        out = CipResult()
        self._chk(self.lib.ssk_da_change_in_pose(self.h, C.byref(this_view), C.byref(other_view), _ptr(m), _ptr(st), _ptr(T),
                                                 C.byref(params), _ptr(res), C.byref(out)))
        return st, res[:n_trk].copy(), out

    # -- B3 ------------------------------------------------------------------------------------------
    def build_observations(self, kp_left, kp_right, matches, lm_params):
        kl = np.ascontiguousarray(kp_left, KEYPOINT_DTYPE)
        kr = np.ascontiguousarray(kp_right, KEYPOINT_DTYPE)
        m = np.ascontiguousarray(matches, DMATCH_DTYPE)
        px = np.zeros((len(m), 4), np.float32)
        pos = np.zeros((len(m), 3), np.float64)
        self._chk(self.lib.ssk_build_observations(self.h, _ptr(kl), _ptr(kr), _ptr(m), len(m), C.byref(lm_params), _ptr(px), _ptr(pos)))
        return px, pos

    def stitch_shared(self, d_gathered, n_ranks, n_shared, d_pos, lam=1e-9):
        """d_pos [n_shared * 3] += (sum_r H_r + lam I)^-1 sum_r g_r over the gathered blocks, rank order."""
        self.lib.ssk_lm_stitch_shared.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
        self._chk(self.lib.ssk_lm_stitch_shared(self.h, _dev(d_gathered), int(n_ranks), int(n_shared), float(lam), _dev(d_pos)))

    def posegraph_optimize(self, X, edge_ij, edge_z, fixed=0, prm=None):
        """ssk_posegraph_optimize: returns (optimised node poses [n, 12], PgResult)."""
        X = np.ascontiguousarray(X, np.float64).reshape(-1, 12).copy()
        ij = np.ascontiguousarray(edge_ij, np.int32).reshape(-1, 2)
        z = np.ascontiguousarray(edge_z, np.float64).reshape(-1, 12)
        prm = prm or PgParams()
        out = PgResult()
        self._chk(self.lib.ssk_posegraph_optimize(self.h, len(X), _ptr(X), len(ij), _ptr(ij), _ptr(z), int(fixed), C.byref(prm), C.byref(out)))
        return X, out

    # -- profiling -------------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self.lib.ssk_profile_enable(self.h, int(on)))

    def profile_read(self):
        n = 14
        ms = (C.c_double * n)()
        calls = (C.c_int64 * n)()
        self._chk(self.lib.ssk_profile_read(self.h, ms, calls))
        self.lib.ssk_stage_name.restype = C.c_char_p
        return {self.lib.ssk_stage_name(i).decode(): (ms[i], calls[i]) for i in range(n)}

    def data_association(self, this_view, other_view, params):
        n = this_view.n_matches
        out = np.zeros(n, DMATCH_DTYPE)
        status = np.zeros(n, np.int32)
        tracked = C.c_int32()
        self._chk(self.lib.ssk_data_association(self.h, C.byref(this_view), C.byref(other_view), C.byref(params),
                                                _ptr(out), _ptr(status), C.byref(tracked)))
        return out, status, tracked.value
