/*
 * ssk_capi.h -- C-ABI drop-in boundary of the MI355X-native srba-stereo-slam hot path.
 *
 * One shared library (libssk_hip.so, built from srba-stereo-slam_amd/csrc with hipcc for gfx950)
 * exports exactly these symbols.  Plain pointers and sizes only; no C++ / torch types.
 * Every entry point cites the reference call it replaces (paths into the upstream tree,
 * famoreno/srba-stereo-slam; EST.cpp = src/CSRBAStereoSLAMEstimator.cpp, KF.h = src/CStereoSLAMKF.h,
 * UTIL.h = src/srba-stereo-slam_utils.h, SRBA.h = src/srba-stereo-slam.h).
 *
 * Conventions
 *   - return value: 0 = SSK_OK, otherwise an ssk_status error code; ssk_last_error(ctx) has the text.
 *     The C++ facade (srba-stereo-slam_amd/host) turns a non-zero code into std::runtime_error, which is
 *     how the reference reports errors (MRPT THROW_EXCEPTION, EST.cpp:288-295).
 *   - "host" entry points take host pointers and copy through the context's staging buffers
 *     (they are the literal drop-ins for the reference call sites, one keyframe at a time);
 *     "_dev" entry points take device pointers (HBM-resident batches, nothing crosses PCIe) and are
 *     asynchronous on the context's HIP stream.
 *   - there is NO CPU fallback anywhere behind this interface: without a usable gfx950 device
 *     ssk_create fails with SSK_ERR_NO_DEVICE and every other call needs a context.
 *   - THREADS: a context serves ONE host thread at a time.  It owns staging state that calls reuse without
 *     locks (the device / pinned arenas, the pinned arena an LM batch's tables are built in, the slab cache
 *     of destroyed LM batches, the error text, the stage profiler).  Concurrent callers take a context
 *     each (contexts are cheap: a stream and lazily grown arenas; bench.py's symbolic pass builds the next
 *     batch on one context while another context's stream solves the current one).  Objects made by one
 *     context may be USED through another (ssk_lm_batch_reset / _run / _results take the context whose
 *     stream they enqueue on; ssk_lm_batch_destroy waits for the last such use before the creating
 *     context's slab cache takes the memory back), but never by two threads at once.
 */
#ifndef SSK_CAPI_H
#define SSK_CAPI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSK_API_VERSION 1

typedef enum ssk_status {
	SSK_OK = 0,
	SSK_ERR_NO_DEVICE = 1,     /* no HIP device / not gfx950 */
	SSK_ERR_BAD_ARG = 2,
	SSK_ERR_HIP = 3,           /* a HIP runtime call failed; see ssk_last_error */
	SSK_ERR_CAPACITY = 4,      /* a caller-provided buffer is too small */
	SSK_ERR_TOO_FEW_MATCHES = 5, /* < 8 stereo matches: reference throws at EST.cpp:288-295 */
	SSK_ERR_NOT_POSDEF = 6     /* reduced camera system not positive definite after max lambda */
} ssk_status;

/* ---- plain records, layout-compatible with the OpenCV 2.4 types the reference stores ----------
 * cv::KeyPoint {Point2f pt; float size, angle, response; int octave, class_id;}  = 28 B (UTIL.h:47)
 * cv::DMatch   {int queryIdx, trainIdx, imgIdx; float distance;}                 = 16 B (UTIL.h:48) */
typedef struct ssk_keypoint { float x, y, size, angle, response; int32_t octave, class_id; } ssk_keypoint;
typedef struct ssk_dmatch { int32_t queryIdx, trainIdx, imgIdx; float distance; } ssk_dmatch;

#define SSK_DESC_BYTES 32   /* ORB descriptor row: 256 bits, CV_8U x 32 (EST.cpp:1351) */

/* per-match status after the data-association filters; values of
 * TVectorDAMatchInfo::TDAMatchStatus (src/CSRBAStereoSLAMEstimator.h:102) */
enum {
	SSK_DA_TRACKED = 0, SSK_DA_NON_TRACKED = 1, SSK_DA_REJ_SLOPE = 2, SSK_DA_REJ_ORB = 3,
	SSK_DA_REJ_FUND_MATRIX = 4, SSK_DA_REJ_CHANGE_POSE = 5, SSK_DA_REJ_CONSISTENCY = 6
};

typedef struct ssk_ctx ssk_ctx;

/* ---- context ----------------------------------------------------------------------------------- */
int ssk_api_version(void);
/* device: HIP ordinal.  hip_stream: a hipStream_t to enqueue on (e.g. torch's current stream), or
 * NULL for a context-owned stream. */
int ssk_create(ssk_ctx** out, int device, void* hip_stream);
void ssk_destroy(ssk_ctx* ctx);
const char* ssk_last_error(const ssk_ctx* ctx);   /* ctx may be NULL: returns the create-time error */
/* The library reads its environment switches (SSK_LM_KERNEL, SSK_LM_SCHUR, ...: DESIGN.md 5 "Switches") ONCE, in ssk_create.  ssk_env_refresh reads them
 * again for this context: a hook for tests and same-box A/Bs that switch kernels on a live context (the Python binding calls it in front of every
 * LM batch create); a drop-in caller never needs it. */
int ssk_env_refresh(ssk_ctx* ctx);
int ssk_synchronize(ssk_ctx* ctx);                /* hipStreamSynchronize on the context stream */
void* ssk_stream(ssk_ctx* ctx);                   /* the hipStream_t kernels are launched on */
/* HIP-event timing on the context stream (bench.py measures kernels on the stream they run on). */
int ssk_timer_start(ssk_ctx* ctx);
int ssk_timer_stop_ms(ssk_ctx* ctx, float* ms);   /* records + synchronises the stop event */

/* Per-stage kernel timing with HIP events on the context stream (one event pair per kernel launch while
 * enabled; adds a small serialisation, so bench.py enables it only for its roofline pass). */
enum {
	SSK_STAGE_FAST = 0,        /* fast_nms_kernel: the HBM-bound image scan */
	SSK_STAGE_SELECT,          /* select_kernel */
	SSK_STAGE_DESCRIBE,        /* describe_kernel */
	SSK_STAGE_STEREO,          /* stereo_match_kernel */
	SSK_STAGE_HAMMING,         /* hamming_match_kernel */
	SSK_STAGE_DA_FILTERS,      /* da_filters_kernel */
	SSK_STAGE_LM_LINEARIZE,    /* residual + Jacobian + Hessian-block build */
	SSK_STAGE_LM_SCHUR,        /* Schur complement */
	SSK_STAGE_LM_SOLVE,        /* dense Cholesky + back-substitution */
	SSK_STAGE_LM_UPDATE,       /* state update + error evaluation */
	SSK_STAGE_CHANGE_IN_POSE,  /* cip_kernel: change-in-pose outlier filter */
	SSK_STAGE_FUND_MATRIX,     /* fm_kernel: fundamental-matrix RANSAC outlier filter */
	SSK_STAGE_BOW,             /* bow_words / bow_vector / bow_score / bow_topk kernels: place recognition */
	SSK_STAGE_POSE_GRAPH,      /* pg_solve_kernel: global pose-graph optimisation */
	SSK_STAGE_COUNT
};
int ssk_profile_enable(ssk_ctx* ctx, int on);     /* also clears the accumulated sums */
/* synchronises the stream, then returns accumulated milliseconds and launch counts per stage */
int ssk_profile_read(ssk_ctx* ctx, double* ms /* SSK_STAGE_COUNT */, int64_t* calls /* SSK_STAGE_COUNT */);
const char* ssk_stage_name(int stage);

/* ---- B2: keyframe <-> keyframe Hamming brute force ---------------------------------------------
 * Replaces  cv::BFMatcher(NORM_HAMMING,false).match(query, train, out)   EST.cpp:1447-1452.
 * out[q] = {queryIdx=q, trainIdx=argmin_t popcount(query[q]^train[t]) (lowest t on ties), imgIdx=0,
 * distance=(float)min}.  nq rows of 32 B, nt rows of 32 B.  nt==0 or nq==0 -> *n_out = 0
 * (BFMatcher clears the output).  out capacity must be >= nq. */
int ssk_hamming_match(ssk_ctx* ctx, const uint8_t* query, int nq, const uint8_t* train, int nt,
                      ssk_dmatch* out, int* n_out);

/* Batched, device-resident form: P independent (query,train) pairs in one launch.
 * d_desc_q / d_desc_t: descriptor pools (rows of 32 B; pool base 16-B aligned).  Pair p matches
 * virtual rows q_off[p]..q_off[p+1] against virtual rows t_off[p]..t_off[p+1] (device int32 arrays of
 * length P+1).  Optional gather (a5, EST.cpp:1354-1358,1435-1439): d_q_gather[r] / d_t_gather[r]
 * (may be NULL = identity) is the pool row that virtual row r reads, so the "matched-only, re-indexed
 * by match" descriptor matrices of the reference are never materialised.
 * Outputs (device int32, indexed by virtual query row): d_train_idx = argmin relative to t_off[p],
 * d_dist = the Hamming distance; (-1, INT32_MAX) when the pair has no train rows.
 * max_query_rows >= max_p (q_off[p+1]-q_off[p]) sizes the grid (offsets live on the device);
 * a pair may hold at most 2^22 train rows. */
int ssk_hamming_match_batched_dev(ssk_ctx* ctx,
                                  const uint8_t* d_desc_q, const int32_t* d_q_off, const int32_t* d_q_gather,
                                  const uint8_t* d_desc_t, const int32_t* d_t_off, const int32_t* d_t_gather,
                                  int n_pairs, int max_query_rows, int32_t* d_train_idx, int32_t* d_dist);

/* ---- B2: data-association filters --------------------------------------------------------------
 * Replaces m_detect_outliers_with_direction (EST.cpp:1883-1946) and
 * m_detect_outliers_with_orb_distance (EST.cpp:1951-2010), applied in that order as
 * m_internal_data_association does (EST.cpp:1482-1515). */
typedef struct ssk_da_params {
	int32_t filter_by_direction;      /* srba_options.da_filter_by_direction (KITTI.ini:124) */
	int32_t filter_by_orb_distance;   /* srba_options.da_filter_by_orb_distance (UTIL.h default true) */
	int32_t rows_offset;              /* stereo_camera.leftCamera.nrows, the 'offset' arg (EST.cpp:1486) */
	float   max_orb_distance_da;      /* srba_options.max_orb_distance_da (KITTI.ini:120) */
} ssk_da_params;

/* One keyframe as the reference stores it (KF.h:99-104), host memory, read-only. */
typedef struct ssk_kf_view {
	const ssk_keypoint* kp_left;  int32_t n_left;
	const ssk_keypoint* kp_right; int32_t n_right;
	const uint8_t* desc_left;     /* n_left  x 32 */
	const uint8_t* desc_right;    /* n_right x 32 */
	const ssk_dmatch* matches;    int32_t n_matches;   /* L<->R matches: queryIdx=left kp, trainIdx=right kp */
} ssk_kf_view;

/* Whole m_internal_data_association up to (not including) the fundamental-matrix / change-in-pose
 * stages (EST.cpp:1413-1515 + output assembly 1647-1669): gathers matched-only left descriptors of
 * both KFs (Q3), brute-force match, direction + ORB-distance filters.
 *   out_matches[k] (k < this.n_matches): the raw BFMatcher result (match-index space)
 *   out_status[k] : SSK_DA_* ;  *out_tracked = number of SSK_DA_TRACKED.
 * Capacity of both output arrays: this_kf->n_matches. */
int ssk_data_association(ssk_ctx* ctx, const ssk_kf_view* this_kf, const ssk_kf_view* other_kf,
                         const ssk_da_params* params, ssk_dmatch* out_matches, int32_t* out_status,
                         int32_t* out_tracked);

/* Filters only, on already-computed matches (host arrays).  this_xy / other_xy: left-keypoint pixel
 * coordinates re-indexed by MATCH index (2 floats per match). status is in/out. */
int ssk_da_filters(ssk_ctx* ctx, const ssk_dmatch* matches, int n_this, int n_other,
                   const float* this_xy, const float* other_xy, const ssk_da_params* params,
                   int32_t* status, int32_t* out_tracked);

/* ---- B2, SURVEY 8(f) row 1: change-in-pose outlier filter -------------------------------------------
 * Replaces m_detect_outliers_with_change_in_pose (EST.cpp:2113-2177), i.e. the tracked-pair marshalling
 * (:2120-2133), stereo-vo's  m_voEngine.getChangeInPose(tracked_pairs, pre/cur matches and features,
 * stereo_camera, odometry_result, initialPoseVector)  (:2139-2146; third-party, absent: a 6-DoF robust
 * Gauss-Newton on the stereo reprojection error of the tracked pairs in two stages, parameters
 * [LEAST_SQUARES] KITTI.ini:40-50) and the residual gate  out_residual[k0] > residual_th  (:2164-2176;
 * residual_th KITTI.ini:119).  Algorithm contract: DESIGN.md "Change-in-pose contract" (P1-P6). */
typedef struct ssk_cip_params {
	double fx_l, fy_l, cx_l, cy_l, fx_r, fy_r, cx_r, cy_r, baseline;   /* srba_options.stereo_camera */
	int32_t initial_max_iters;   /* KITTI.ini:41  iterations of the first stage (all tracked pairs) */
	int32_t max_iters;           /* KITTI.ini:42  iterations of the refinement stage (first-stage inliers) */
	int32_t max_incr_cost;       /* KITTI.ini:43  times the cost may grow before a stage stops */
	int32_t use_robust_kernel;   /* KITTI.ini:49 */
	double residual_threshold;   /* KITTI.ini:44  inlier gate between the stages (squared pixel norm) */
	double min_mod_out_vector;   /* KITTI.ini:45  a stage stops when the step norm falls below this */
	double kernel_param;         /* KITTI.ini:50  pseudo-Huber parameter */
	int32_t bad_tracking_th;     /* KITTI.ini:47  fewer usable pairs than this: the estimate is invalid */
	int32_t reserved;
	double residual_th;          /* KITTI.ini:119 final gate: larger residual -> SSK_DA_REJ_CHANGE_POSE */
} ssk_cip_params;

typedef struct ssk_cip_result {
	double pose[12];             /* row-major 3x4 [R|t]: the OTHER (previous) frame seen from THIS one (odometry_result.outPose) */
	double cost;                 /* robustified cost of the second-stage inliers at the final pose */
	int32_t valid;               /* odometry_result.valid */
	int32_t iters_initial, iters_final;   /* odometry_result.num_it, num_it_final */
	int32_t n_pairs, n_inliers, n_rejected;
} ssk_cip_result;

/* Host drop-in of m_detect_outliers_with_change_in_pose for one keyframe pair.  da_matches[k].trainIdx is the
 * OTHER keyframe's match tracked by match k of THIS keyframe (this_matches.m_matches), status is
 * this_matches.m_status (in/out).  init_pose: [R|t] 3x4 of the previous frame wrt the current one (the caller
 * converts kf_ini_rel_pose, EST.cpp:2135-2137).  residuals (optional): out_residual per tracked pair, in order. */
int ssk_da_change_in_pose(ssk_ctx* ctx, const ssk_kf_view* this_kf, const ssk_kf_view* other_kf,
                          const ssk_dmatch* da_matches, int32_t* status, const double init_pose[12],
                          const ssk_cip_params* params, double* residuals, ssk_cip_result* out);

/* Known-geometry entry (what the kernel consumes; tests and the batched path use it): per match k of THIS
 * keyframe the stereo observation this_obs[4k..] = (xl, yl, xr, yr), per match j of the OTHER keyframe
 * other_obs[4j..]; train_idx / status as above (host arrays). */
int ssk_cip_filter(ssk_ctx* ctx, const float* this_obs, int n_this, const float* other_obs, int n_other,
                   const int32_t* train_idx, int32_t* status, const double init_pose[12],
                   const ssk_cip_params* params, double* residuals, ssk_cip_result* out);

/* ---- B2, SURVEY 8(f) row 3: fundamental-matrix outlier filter ------------------------------------------
 * Replaces m_detect_outliers_with_F (EST.cpp:2015-2055): left keypoints of the tracked pairs (:2022-2040),
 * cv::findFundamentalMat(p_other_left, p_this_left, FM_RANSAC, max_y_diff_epipolar, ransac_fit_prob, mask) (:2044;
 * OpenCV 2.4, absent: 7-point RANSAC with adaptive iteration count) and the verdict mask 0 => sREJ_FUND_MATRIX
 * (:2047-2054).  OpenCV samples from its global RNG, so parity with the reference is statistical by construction;
 * this build's generator is counter based (contract F1-F6, DESIGN.md "Fundamental-matrix contract"). */
typedef struct ssk_fm_params {
	double threshold;            /* srba_options.max_y_diff_epipolar (KITTI.ini:121): max distance to the epipolar line, px */
	double confidence;           /* srba_options.ransac_fit_prob (KITTI.ini:122) */
	double norm_cx, norm_cy, norm_scale;   /* fixed normalisation of the pixel coordinates inside the 7-point solver */
	uint64_t seed;               /* of the counter-based sample generator */
	int32_t max_iters;           /* OpenCV: 2000 */
	int32_t min_points;          /* 15: below it OpenCV 2.4 would switch to LMedS and the reference does not call (EST.cpp:1590) */
} ssk_fm_params;

typedef struct ssk_fm_result {
	double F[9];                 /* row-major, m2^T F m1 = 0 with m1 = other keyframe, unit Frobenius norm */
	int32_t valid, n_pairs, n_inliers, n_rejected;
	int32_t hypotheses;          /* samples drawn before the adaptive stop */
	int32_t best_hypothesis;
} ssk_fm_result;

/* Host drop-in of m_detect_outliers_with_F for one keyframe pair (arguments as ssk_da_change_in_pose). */
int ssk_da_fundamental_filter(ssk_ctx* ctx, const ssk_kf_view* this_kf, const ssk_kf_view* other_kf,
                              const ssk_dmatch* da_matches, int32_t* status, const ssk_fm_params* params, ssk_fm_result* out);

/* Known-geometry entry: this_xy[2k..] / other_xy[2j..] = left keypoint pixel of match k / j (host arrays). */
int ssk_fm_filter(ssk_ctx* ctx, const float* this_xy, int n_this, const float* other_xy, int n_other,
                  const int32_t* train_idx, int32_t* status, const ssk_fm_params* params, ssk_fm_result* out);

/* The three stages of m_internal_data_association for one keyframe pair behind ONE upload, one synchronisation and one
 * download: ssk_data_association, then (fm_params != NULL) the fundamental-matrix filter, then (cip_params != NULL) the
 * change-in-pose filter from init_pose (NULL: identity), each on the statuses the previous one left.  The host gates between
 * the stages ("not enough input data", EST.cpp:1590-1594, 1617-1621) are NOT applied: all requested stages run, and `out`
 * carries the number of tracked matches after each so that the caller applies the gates as the reference does (a stage run
 * on fewer than its minimum leaves the statuses alone).  Same kernels, same inputs, same results as the three calls. */
typedef struct ssk_da_staged_out {
	int32_t tracked_da, tracked_fm, tracked_final, reserved;   /* SSK_DA_TRACKED count after each stage (a skipped stage repeats the previous) */
	ssk_fm_result fm;
	ssk_cip_result cip;
} ssk_da_staged_out;
int ssk_data_association_staged(ssk_ctx* ctx, const ssk_kf_view* this_kf, const ssk_kf_view* other_kf, const ssk_da_params* da_params,
                                const ssk_fm_params* fm_params, const ssk_cip_params* cip_params, const double* init_pose,
                                ssk_dmatch* out_matches, int32_t* out_status, ssk_da_staged_out* out);
/* The same for ONE keyframe against n_others keyframes (m_data_association, EST.cpp:1341-1411: the candidate against every
 * similar keyframe) in one upload, one launch per stage and one synchronisation.  Pair p = (this_kf, others[p]); fm_seeds:
 * RANSAC seed per pair or NULL (fm_params->seed + p); init_poses: n_others x 12 or NULL (identity);
 * out_matches / out_status: n_others rows of this_kf->n_matches entries; outs: n_others records. */
int ssk_data_association_staged_multi(ssk_ctx* ctx, const ssk_kf_view* this_kf, const ssk_kf_view* others, int n_others,
                                      const ssk_da_params* da_params, const ssk_fm_params* fm_params, const uint64_t* fm_seeds,
                                      const ssk_cip_params* cip_params, const double* init_poses,
                                      ssk_dmatch* out_matches, int32_t* out_status, ssk_da_staged_out* outs);

/* ---- B1: stereo front-end ------------------------------------------------------------------------
 * Supplies the definition CStereoSLAMKF::create() lacks (declared KF.h:82-86, defined nowhere) and
 * replaces the live equivalent  m_voEngine.processNewImagePair() + getValues()
 * (EST.cpp:112,267; KF.h:68-70), i.e. stereo-vo's ORB detect/describe (via cv::ORB) and its
 * "descriptor row-by-row" L<->R matcher (EST.cpp:1135-1137).  Algorithm contract: DESIGN.md
 * "Front-end contract"; parameters and their reference origin: */
typedef struct ssk_frontend_params {
	int32_t n_feats;          /* ORB nfeatures: srba_options.n_feats -> orb_nfeats (EST.cpp:1142, KITTI.ini:112); <= 2048 */
	int32_t fast_threshold;   /* FAST threshold: detect_fast_th -> initial_FAST_threshold (EST.cpp:1140, KITTI.ini:114) */
	int32_t n_levels;         /* must be 1: n_levels -> orb_nlevels (EST.cpp:1141, UTIL.h:288, KITTI.ini:5) */
	int32_t nms_radius;       /* keypoint non-max suppression half window; 2 = the "5x5 windows" of KITTI.ini:12; 0 = off */
	int32_t orb_max_distance; /* max Hamming distance of an accepted L<->R match (KITTI.ini:24) */
	float   max_y_diff;       /* epipolar band |yL-yR| <= max_y_diff (KITTI.ini:17) */
	float   min_disparity;    /* accepted iff min_disparity <= xL-xR <= max_disparity (positive depth) */
	float   max_disparity;
	float   harris_k;         /* 0.04: cv::ORB HARRIS_SCORE */
} ssk_frontend_params;
/* KITTI.ini values: {500, 20, 1, 2, 60, 2.0f, 1.0f, 128.0f, 0.04f}  (disparity bound: depth >= 3 m at fx*b = 382) */

/* One keyframe out, caller-allocated arrays of `capacity` (>= n_feats) records each (a1: KF.h:99-104). */
typedef struct ssk_kf_out {
	int32_t capacity;
	ssk_keypoint* kp_left;  ssk_keypoint* kp_right;
	uint8_t* desc_left;     uint8_t* desc_right;      /* capacity x 32 */
	ssk_dmatch* matches;                              /* queryIdx = left kp, trainIdx = right kp */
	int32_t n_left, n_right, n_matches;               /* filled by the call */
} ssk_kf_out;

/* BRIEF sampling pattern: 256 tests x (x0,y0,x1,y1) int8, |v| <= 13.  Default: include/ssk_brief_pattern.h. */
int ssk_set_brief_pattern(ssk_ctx* ctx, const int8_t* pattern1024);
int ssk_get_brief_pattern(ssk_ctx* ctx, int8_t* pattern1024);
/* A context starts with the pattern in the 1024-byte file named by the environment variable SSK_BRIEF_PATTERN_FILE when it is set (e.g.
 * OpenCV's learned bit_pattern_31_, which this repository cannot ship: needed for descriptors that match a DBoW2 vocabulary trained on
 * cv::ORB); a file that cannot be read, is short or holds a coordinate beyond +-13 fails ssk_create. */
/* Steering of the BRIEF tests (front-end contract C7): 1 (default) = as cv::ORB -- (float)cos / (float)sin of the fastAtan2 angle times
 * (float)(CV_PI/180.f), include/ssk_sincos.h; 0 = the exact direction of the intensity centroid (the form of rounds 1-2). */
int ssk_set_orb_steering(ssk_ctx* ctx, int mode);

/* Host drop-in, one stereo pair (8-bit gray, `stride` bytes between rows):
 * detect + describe both images, match L<->R, fill `out`.  Does NOT throw on few matches: the
 * adaptive-threshold loop and the "< 8 matches" exception (EST.cpp:271-315) live in the C++ facade. */
int ssk_frontend_run(ssk_ctx* ctx, const uint8_t* left, const uint8_t* right, int width, int height,
                     int stride, const ssk_frontend_params* params, ssk_kf_out* out);

/* Device-resident batch of keyframes (images, keypoints, descriptors, matches all stay in HBM). */
typedef struct ssk_kfbatch ssk_kfbatch;
int ssk_kfbatch_create(ssk_ctx* ctx, int n_kf, int width, int height, int max_feats, ssk_kfbatch** out);
void ssk_kfbatch_destroy(ssk_kfbatch* b);
/* Device image storage: pixel (x,y) of eye e (0 left, 1 right) of keyframe k is at
 * d_images[((k*2+e)*height + y)*pitch + x].  Zeroed and idle when ssk_kfbatch_create returns: the caller may fill it on any stream, and
 * makes that stream's writes visible to the context's stream (an event or a synchronisation) before ssk_frontend_run_batch. */
int ssk_kfbatch_images(ssk_kfbatch* b, uint8_t** d_images, int* pitch);
int ssk_kfbatch_upload(ssk_ctx* ctx, ssk_kfbatch* b, int kf, const uint8_t* left, const uint8_t* right, int stride);
/* Front-end over keyframes [kf_begin, kf_end) of the batch, asynchronous on the context stream. */
int ssk_frontend_run_batch(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, const ssk_frontend_params* params);
/* Synchronises, then packs keyframe `kf` into the reference's AoS records. */
int ssk_kfbatch_download(ssk_ctx* ctx, ssk_kfbatch* b, int kf, ssk_kf_out* out);
/* Device-side counts (int32 [n_kf*2] keypoints per image, [n_kf] matches), for benchmarks / checksums. */
int ssk_kfbatch_counts(ssk_ctx* ctx, ssk_kfbatch* b, int32_t* h_n_kp /* n_kf*2 */, int32_t* h_n_matches /* n_kf */);
/* Debug/test hook: FAST+NMS candidates of image (kf, eye) as packed keys
 * (score<<24 | (4095-y)<<12 | (4095-x)), sorted descending. Returns the count in *n (may exceed cap). */
int ssk_kfbatch_debug_candidates(ssk_ctx* ctx, ssk_kfbatch* b, int kf, int eye, uint32_t* keys, int cap, int* n);

/* Batched data association between keyframes of two device-resident batches (B2 on HBM-resident data):
 * pair p associates keyframe pairs[2p] of `this_b` with keyframe pairs[2p+1] of `other_b` (host int32
 * array).  Outputs are device int32 arrays, row p holds max_feats entries (only the first n_matches of
 * the "this" keyframe are meaningful): d_train_idx, d_dist, d_status [n_pairs*max_feats], d_tracked [n_pairs]. */
int ssk_da_run_batch(ssk_ctx* ctx, ssk_kfbatch* this_b, ssk_kfbatch* other_b, const int32_t* pairs, int n_pairs,
                     const ssk_da_params* params, int32_t* d_train_idx, int32_t* d_dist, int32_t* d_status,
                     int32_t* d_tracked);

/* Batched change-in-pose filter on the outputs of ssk_da_run_batch (same pairs, same device arrays): d_status is
 * updated in place, d_tracked recounted.  init_poses: host [n_pairs*12] or NULL (identity).  d_results: device
 * array of n_pairs ssk_cip_result or NULL.  Asynchronous on the context stream. */
int ssk_cip_run_batch(ssk_ctx* ctx, ssk_kfbatch* this_b, ssk_kfbatch* other_b, const int32_t* pairs, int n_pairs,
                      const ssk_cip_params* params, const double* init_poses, const int32_t* d_train_idx,
                      int32_t* d_status, int32_t* d_tracked, ssk_cip_result* d_results);

/* Batched fundamental-matrix filter on the outputs of ssk_da_run_batch (same conventions as ssk_cip_run_batch);
 * pair p uses seed params->seed + p. */
int ssk_fm_run_batch(ssk_ctx* ctx, ssk_kfbatch* this_b, ssk_kfbatch* other_b, const int32_t* pairs, int n_pairs,
                     const ssk_fm_params* params, const int32_t* d_train_idx, int32_t* d_status, int32_t* d_tracked,
                     ssk_fm_result* d_results);

/* ---- bag-of-words place recognition: CBoWManager (src/CBoWManager.h:59-109) ---------------------------------
 * Replaces DBoW2 1.0.2's BriefVocabulary::transform and BriefDatabase::add / ::query for the vocabulary the reference
 * loads (demo/voc.yml.gz: k = 8, L = 5, weightingType 0 = TF_IDF, scoringType 0 = L1_NORM; any k, L and tree shape
 * are accepted, other weighting / scoring types are refused).  Contract W1-W5: oracle/oracle_bow.c, DESIGN.md.
 *   m_voc.load(file); m_db.setVocabulary(m_voc, true, 5)  (BOW.h:59-66)  ->  ssk_bow_vocab_create + ssk_bow_db_create
 *   m_db.add(features)            (BOW.h:71-77, EST.cpp:118,919)         ->  ssk_bow_transform + ssk_bow_db_add
 *   m_db.query(features, ret, 4)  (BOW.h:84-90, EST.cpp:426)             ->  ssk_bow_transform + ssk_bow_db_query
 * The file parser (OpenCV FileStorage YAML, gzip) lives in the C++ facade (host/CBoWManager.h); this interface takes the
 * vocabulary as flat arrays.  Entry ids are 0, 1, 2 ... in insertion order, which the reference reads as keyframe ids. */
#define SSK_BOW_MAX_RESULTS 16
typedef struct ssk_bow_vocab ssk_bow_vocab;
typedef struct ssk_bow_db ssk_bow_db;
/* n_rec node records IN FILE ORDER (the root, node 0, has no record): node_id > 0, parent_id >= 0, weight, 32-byte
 * descriptor with bit m + 8k of DBoW2's bitset in bit m of byte k (BOW.h:95-107).  word_node[w] = nodeId of word w. */
int ssk_bow_vocab_create(ssk_ctx* ctx, int k, int L, int weighting, int scoring, int n_rec, const int32_t* node_id,
                         const int32_t* parent_id, const double* weight, const uint8_t* desc, int n_words,
                         const int32_t* word_node, ssk_bow_vocab** out);
void ssk_bow_vocab_destroy(ssk_bow_vocab* v);
int ssk_bow_vocab_info(const ssk_bow_vocab* v, int* k, int* L, int* n_nodes, int* n_words);
/* Host drop-in, one keyframe: n descriptors (rows of m_descriptors_left) -> bag-of-words vector, ascending word ids
 * and L1-normalised values; words / values have capacity n. */
int ssk_bow_transform(ssk_ctx* ctx, const ssk_bow_vocab* v, const uint8_t* desc, int n, int32_t* words, double* values,
                      int* n_words);
/* Batched, device resident: left descriptors of keyframes [kf_begin, kf_end) of the batch -> the batch's bag-of-words
 * arrays (asynchronous on the context stream). */
int ssk_bow_transform_batch(ssk_ctx* ctx, const ssk_bow_vocab* v, ssk_kfbatch* b, int kf_begin, int kf_end);
int ssk_kfbatch_bow_download(ssk_ctx* ctx, ssk_kfbatch* b, int kf, int32_t* words, double* values, int cap, int* n);
/* Database of at most capacity_entries vectors of at most max_words_per_entry words each (HBM resident). */
int ssk_bow_db_create(ssk_ctx* ctx, const ssk_bow_vocab* v, int capacity_entries, int max_words_per_entry, ssk_bow_db** out);
void ssk_bow_db_destroy(ssk_bow_db* d);
int ssk_bow_db_size(const ssk_bow_db* d);
int ssk_bow_db_clear(ssk_ctx* ctx, ssk_bow_db* d);
int ssk_bow_db_add(ssk_ctx* ctx, ssk_bow_db* d, const int32_t* words, const double* values, int n, int* entry_id);
int ssk_bow_db_add_from_batch(ssk_ctx* ctx, ssk_bow_db* d, ssk_kfbatch* b, int kf, int* entry_id);   /* device to device */
/* L1 query: up to max_results (1..SSK_BOW_MAX_RESULTS) entries by descending score, ties by ascending entry id; entries
 * with id >= max_id are skipped unless max_id == -1; entries without a common word are not reported. */
int ssk_bow_db_query(ssk_ctx* ctx, ssk_bow_db* d, const int32_t* words, const double* values, int n, int max_results, int max_id,
                     int32_t* ids, double* scores, int* n_results);
/* Batched: the bag-of-words vectors of keyframes [kf_begin, kf_end) of the batch against d; max_ids: host array, one per
 * query, or NULL (-1).  Results stay in the batch (asynchronous); read them with ssk_kfbatch_bow_results, which writes
 * SSK_BOW_MAX_RESULTS slots per query. */
int ssk_bow_db_query_batch(ssk_ctx* ctx, ssk_bow_db* d, ssk_kfbatch* b, int kf_begin, int kf_end, int max_results,
                           const int32_t* max_ids);
int ssk_kfbatch_bow_results(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, int32_t* ids, double* scores,
                            int32_t* n_results);

/* ---- global pose-graph optimisation after the frame loop (last item) ------------------------
 * Replaces  rba.get_global_graphslam_problem(poseGraph); mrpt::graphslam::optimize_graph_spa_levmarq(poseGraph, ...)
 * (EST.cpp:945-957).  Contract G1-G6: oracle/oracle_posegraph.c.  Nodes X_k: pose of keyframe k in the root's frame,
 * row-major [R|t], in/out; edge e: (i, j) = edge_ij[2e], edge_ij[2e+1] with X_i^-1 X_j = Z_e (edge_z, 12 doubles each; the
 * k2k edge {from, to, inv_pose} gives i = to, j = from, Z = inv_pose); fixed_node is held (the root).  Levenberg-Marquardt on
 * e = (t_E, log R_E), E = Z^-1 X_i^-1 X_j, unit information, gain-ratio lambda update (as MRPT's optimiser).  The step
 * (H + lambda I) d = -g is solved either directly -- block elimination on the spanning tree (no fill) plus a Woodbury
 * correction of rank 6 per edge off the tree: a keyframe graph is a tree plus a few loop closures -- or, when more than
 * SSK_PG_DIRECT_MAX_EDGES edges lie off the tree, by conjugate gradients preconditioned with the spanning-tree part (the
 * oracle's solver; `solver` = 1 forces it). */
#define SSK_PG_DIRECT_MAX_EDGES 42
typedef struct ssk_pg_params {
	int32_t max_iters;                 /* LM attempts; MRPT's default is 100 */
	int32_t pcg_max_iters;             /* CG iterations per attempt */
	double tau;                        /* lambda0 = tau * max diag(H); 1e-3 (MRPT's tau) */
	double max_lambda;                 /* 1e20 */
	double converged_rel_decrease;     /* stop when an attempt changes the cost by <= this fraction; 1e-9 */
	double pcg_tol;                    /* CG stops at r.z <= pcg_tol^2 r0.z0; 1e-10 */
	int32_t solver;                    /* 0: direct when at most SSK_PG_DIRECT_MAX_EDGES edges lie off the spanning tree, else CG;
	                                      1: CG always; 2: direct, SSK_ERR_CAPACITY when it does not apply */
	int32_t reserved;                  /* 0 */
} ssk_pg_params;
typedef struct ssk_pg_result {
	int32_t iters, accepted;
	int32_t stop_reason;               /* 0 max_iters, 1 cost at round-off level, 2 max_lambda, 3 nothing to optimise, 4 converged */
	int32_t pcg_iters_total;           /* 0 when every attempt used the direct solver */
	double cost_init, cost_final, lambda_final;
} ssk_pg_result;
int ssk_posegraph_optimize(ssk_ctx* ctx, int n_nodes, double* node_poses, int n_edges, const int32_t* edge_ij,
                           const double* edge_z, int fixed_node, const ssk_pg_params* params, ssk_pg_result* out);

/* ---- B3: relative bundle adjustment, Levenberg-Marquardt with Schur complement + dense Cholesky ----
 * Replaces the numerical core of  rba.define_new_keyframe(listObs, newKFInfo, true)  (EST.cpp:782-784):
 * SRBA's optimize_edges / optimize_local_area for the problem types of SRBA.h:30-54
 * (kf2kf_poses::SE3, landmarks::Euclidean3D, observations::StereoCamera, sensor_pose_on_robot_se3,
 * solver_LM_schur_dense_cholesky).  One "window" = the k2k edges and landmarks optimised by one call plus
 * everything their observations' spanning-tree paths reference.  Algorithm contract: DESIGN.md "LM contract".
 *
 * Conventions: an edge's inv_pose is the pose of its `from` keyframe seen from its `to` keyframe (SRBA's
 * k2k_edge_t::inv_pose), stored as 12 doubles = row-major 3x4 [R|t]; it maps a point in the `from` frame to
 * the `to` frame.  A landmark position is relative to its base keyframe.  An observation's path lists the
 * edges from the OBSERVER keyframe towards the landmark's BASE keyframe; entry = edge*2 + dir: dir 0 applies
 * inv_pose (the next keyframe on the path is the edge's `from`), dir 1 applies its inverse (the next keyframe
 * is the edge's `to`).  An empty path = observed from its base keyframe. */
typedef struct ssk_lm_window {
	int32_t n_edges_total;             /* edges referenced by any path (optimised + fixed) */
	const int32_t* edge_unknown_slot;  /* [n_edges_total] 0..E-1 for optimised edges (each slot once), -1 = fixed */
	const double*  edge_inv_pose;      /* [n_edges_total][12] */
	int32_t n_lms_total;
	const int32_t* lm_unknown_slot;    /* [n_lms_total] 0..L-1 for optimised landmarks, -1 = fixed */
	const double*  lm_pos;             /* [n_lms_total][3] */
	int32_t n_obs;
	const int32_t* obs_lm;             /* [n_obs] landmark index */
	const float*   obs_px;             /* [n_obs][4] left x,y, right x,y (TPixelCoordf, EST.cpp:694-695) */
	const int32_t* obs_path_off;       /* [n_obs+1] */
	const int32_t* path_edge;          /* [obs_path_off[n_obs]] edge*2+dir */
} ssk_lm_window;

typedef struct ssk_lm_params {
	double fx_l, fy_l, cx_l, cy_l, fx_r, fy_r, cx_r, cy_r, baseline;   /* srba_options.stereo_camera (UTIL.h:366) */
	double sensor_pose[12];            /* camera on robot, [R|t] (EST.cpp:1105-1112; sensor_pose_on_robot_se3) */
	int32_t max_iters;                 /* SRBA parameters.srba.max_iters, default 20 */
	int32_t use_robust_kernel;         /* srba_use_robust_kernel (KITTI.ini:107) */
	double kernel_param;               /* pseudo-Huber parameter (KITTI.ini:109: 1.5) */
	double tau;                        /* initial lambda = tau * max diag(H); 1e-3 */
	double max_lambda;                 /* 1e20 */
	double max_error_per_obs_to_stop;  /* stop when sqrt(err/n_obs) falls below this; 1e-9 */
	double converged_rel_decrease;     /* stop when an attempt changes the error by <= this fraction (either sign);
	                                      1e-9; 0 disables.  The step is kept iff it lowered the error. */
} ssk_lm_params;

typedef struct ssk_lm_result {
	int32_t iters;                     /* solve attempts made (accepted + rejected) */
	int32_t accepted;                  /* accepted steps */
	int32_t stop_reason;               /* 0 max_iters, 1 error threshold, 2 max_lambda, 3 nothing to optimise, 4 converged,
	                                      5 cooperative kernel aborted (barrier timeout; ssk_lm_batch_results reports an error) */
	int32_t n_obs;
	double err_init, err_final;        /* total (robustified) squared error: optimize_results.total_sqr_error_* (EST.cpp:179) */
	double lambda_final;
} ssk_lm_result;

/* one attempt of the LM loop, recorded when the batch was created with trace != 0 */
typedef struct ssk_lm_trace_entry {
	double lambda, err_before, err_after, delta_p_norm, delta_f_norm;
	int32_t accepted, cholesky_ok;
} ssk_lm_trace_entry;

typedef struct ssk_lm_batch ssk_lm_batch;
/* Uploads n windows (state + symbolic structure) to HBM.  trace != 0 additionally keeps, per attempt, the
 * trace entry and the full update vectors (tests).  The caller's arrays are read before the call returns and may be freed then.  A large batch's
 * tables may still be on their way to the device at that point (queued on the context's stream from the library's own pinned memory; every use of
 * the batch, through any context, is ordered behind them): the context's next ssk_lm_batch_create waits for them before it reuses that memory, and
 * ssk_lm_batch_destroy waits for whatever was queued on the batch.  SSK_LM_SYNC_UPLOAD=1: the create waits itself (the form up to round 5). */
int ssk_lm_batch_create(ssk_ctx* ctx, const ssk_lm_window* windows, int n_windows, int trace, ssk_lm_batch** out);
void ssk_lm_batch_destroy(ssk_lm_batch* b);
/* Restores every window's state to what was uploaded (so a benchmark can repeat the same solve). Async. */
int ssk_lm_batch_reset(ssk_ctx* ctx, ssk_lm_batch* b);
/* Runs the LM loop of every window of the batch (one launch, asynchronous on the context stream). */
int ssk_lm_batch_run(ssk_ctx* ctx, ssk_lm_batch* b, const ssk_lm_params* params);
int ssk_lm_batch_results(ssk_ctx* ctx, ssk_lm_batch* b, ssk_lm_result* out /* n_windows */);
/* Name of the kernel ssk_lm_batch_run launches for this batch (diagnostics: the profile summaries under profiles/ are keyed by it). */
int ssk_lm_batch_kernel_name(ssk_ctx* ctx, ssk_lm_batch* b, char* out, int cap);
/* Host time ssk_lm_batch_create spent on the batch's symbolic structure (suffix tries, landmark groups, block-sorted items: what SRBA's
 * define_new_keyframe does on the CPU before it optimises, EST.cpp:782-784), milliseconds, and the host threads it used.  Either may be NULL. */
int ssk_lm_batch_symbolic_ms(ssk_ctx* ctx, ssk_lm_batch* b, double* ms, int* threads);
/* the dense solver of the reduced camera system the batch's heaviest kernel family uses (solver_LM_schur_dense_cholesky, SRBA.h:34): diagnostics / tests */
int ssk_lm_batch_solver_name(ssk_ctx* ctx, ssk_lm_batch* b, char* out, int cap);
/* Current state of window w: edge_inv_pose [n_edges_total*12], lm_pos [n_lms_total*3]. */
int ssk_lm_batch_get_state(ssk_ctx* ctx, ssk_lm_batch* b, int w, double* edge_inv_pose, double* lm_pos);
/* Replaces the state of window w (either pointer may be NULL: that part is kept) -- the structure of a window can be built before its
   initial values are final: the estimator's facade builds the local-area window of a keyframe beside the solve of its new edges
   (the reference runs the two solves back to back on one set of structures: rba.define_new_keyframe, EST.cpp:782-784).  Synchronous. */
/* the state of EVERY window with one download per family: the windows' edge poses (12 doubles x n_edges_total each) and landmarks (3 doubles x
 * n_lms_total each) back to back in window order -- what a caller of a 1 024-window batch wants instead of 1 024 ssk_lm_batch_get_state calls
 * (ssk_rba_define_kf_batch, include/ssk_rba.h).  Either pointer may be NULL. */
int ssk_lm_batch_get_state_all(ssk_ctx* ctx, ssk_lm_batch* b, double* edge_inv_pose, double* lm_pos);
int ssk_lm_batch_set_state(ssk_ctx* ctx, ssk_lm_batch* b, int w, const double* edge_inv_pose, const double* lm_pos);
/* Trace of window w: entries [max_iters]; delta_p [max_iters][6E], delta_f [max_iters][3L] (may be NULL). */
int ssk_lm_batch_get_trace(ssk_ctx* ctx, ssk_lm_batch* b, int w, ssk_lm_trace_entry* entries, int cap,
                           double* delta_p, double* delta_f, int* n_entries);

/* Multi-GPU exchange (BASELINE configs[4]; no reference counterpart, the reference is single-process): submaps are
 * partitioned one per GPU and stitched through landmarks of ONE identity observed from more than one submap.  Such a landmark is
 * based at a keyframe several submaps reach and is optimised by NO window (lm_unknown_slot -1); the exchange moves it:
 * pack   : per shared landmark s (d_lm_idx[s] = its index in window w, or -1 when this window does not observe it) 12 doubles:
 *          position (3), g = sum Jf^T w r (3), upper triangle of H = sum Jf^T Jf (6) over the window's observations of it at the
 *          window's CURRENT state (after ssk_lm_batch_run: the solved state) -- the payload of the ONE all-gather per step (RCCL,
 *          issued by the host layer on the same stream).
 * stitch : the gathered blocks of n_ranks GPUs ([n_ranks][n_shared][12], rank-major as all_gather lays them out) summed in rank order,
 *          so every GPU computes bit-identical results; Gauss-Newton step of every landmark with the cameras held,
 *          d_pos[s] += (sum_r H_r + lambda I)^-1 sum_r g_r  (d_pos: [n_shared*3], the consensus positions every rank holds).
 * set    : d_pos back into window w (current AND initial state, so that ssk_lm_batch_reset keeps them).
 * Alternated with the windows' own solves this is block-coordinate descent on the joint problem of all submaps: it converges to the
 * solution a single process would compute for their union (tests/test_shared_gpu.py, tests/test_dist_cpu.py). */
int ssk_lm_batch_pack_shared(ssk_ctx* ctx, ssk_lm_batch* b, int w, int n_shared, const int32_t* d_lm_idx, const ssk_lm_params* p, double* d_out);
int ssk_lm_stitch_shared(ssk_ctx* ctx, const double* d_gathered, int n_ranks, int n_shared, double lambda, double* d_pos);
/* exchange: the all-gather itself for callers that hold an RCCL communicator (ncclComm_t, passed as void*) instead of torch.distributed:
 *          ncclAllGather of this rank's n_shared x 12 doubles (d_shared, from pack) into d_gathered ([n_ranks][n_shared][12]) on the context's
 *          stream -- behind the pack kernel, ahead of the stitch kernel, no host synchronisation.  RCCL is resolved at run time (dlopen). */
int ssk_lm_exchange_shared(ssk_ctx* ctx, void* nccl_comm, const double* d_shared, double* d_gathered, int n_shared);
int ssk_lm_batch_set_shared(ssk_ctx* ctx, ssk_lm_batch* b, int w, int n_shared, const int32_t* d_lm_idx, const double* d_pos);
/* The blocks H_ff / g_f of the landmarks a window DOES optimise stay on chip; on != 0 before ssk_lm_batch_run leaves those of the last
 * linearisation in HBM (diagnostics). */
int ssk_lm_batch_set_export(ssk_ctx* ctx, ssk_lm_batch* b, int on);

/* a10/a11: observation marshalling of EST.cpp:142-161,667-704: per match, feat id + pixels + the initial
 * landmark position  camera_pose_on_robot (+) projectMatchTo3D(xl,yl,xr)  (UTIL.h:558-574). Host arrays. */
int ssk_build_observations(ssk_ctx* ctx, const ssk_keypoint* kp_left, const ssk_keypoint* kp_right,
                           const ssk_dmatch* matches, int n_matches, const ssk_lm_params* cam,
                           float* obs_px /* n x 4 */, double* init_pos /* n x 3 */);

/* The same for keyframes [kf_begin, kf_end) of a resident batch after ssk_kfbatch_run_frontend, device to device: d_obs_px [n_kf][max_feats][4]
 * (xl, yl, xr, yr of every L<->R match, match order), d_init_pos [n_kf][max_feats][3]; slots beyond a keyframe's match count are left as they
 * are.  Either output may be NULL.  Asynchronous on the context stream. */
int ssk_build_observations_batch(ssk_ctx* ctx, ssk_kfbatch* b, int kf_begin, int kf_end, const ssk_lm_params* cam,
                                 float* d_obs_px, double* d_init_pos);

/* Test hook: the 19 float thresholds the context derived from the host libm's atanf for the
 * direction bins 9..27 (see csrc/ssk_context.hip). */
int ssk_debug_direction_thresholds(ssk_ctx* ctx, float* out19);

#ifdef __cplusplus
}
#endif
#endif /* SSK_CAPI_H */
