/*
 * ssk_rba.h -- C-ABI of the relative-bundle-adjustment ENGINE behind seam B3: the calls the reference's estimator makes on
 * its `rba` member (mySRBA : RbaEngine<...>, src/srba-stereo-slam.h:30-310), for callers that cannot include the C++ facade.
 * Exported by libssk_facade.so (srba-stereo-slam_amd/host, built over libssk_hip.so; same conventions as ssk_capi.h: 0 = SSK_OK,
 * otherwise an ssk_status code and ssk_rba_last_error has the text).  Graph bookkeeping (keyframes, k2k edges, landmarks, the
 * fixed-size-submap edge_creation_policy of SRBA.h:144-308, spanning trees) is host code, as in SRBA itself; every
 * Levenberg-Marquardt solve runs on the MI355X through ssk_lm_batch_* (there is no CPU solver behind this interface).
 *
 *   rba.parameters.* = ...                                   (EST.cpp:1149-1160)   ->  ssk_rba_create(options)
 *   rba.setInitialKFPose(pose)                               (EST.cpp:772-778)     ->  ssk_rba_set_initial_kf_pose
 *   rba.define_new_keyframe(listObs, newKFInfo, optimize)    (EST.cpp:166,782)     ->  ssk_rba_define_kf
 *   rba.get_rba_state().k2k_edges[i] / .unknown_lms          (EST.cpp:1779, SRBA.h)->  ssk_rba_get_edge / ssk_rba_get_landmark
 *   rba.create_complete_spanning_tree(root, tree, depth)     (EST.cpp:909-915)     ->  ssk_rba_spanning_tree_pose
 *   rba.get_global_graphslam_problem + optimize_graph_spa_levmarq (EST.cpp:945-957)->  ssk_rba_global_poses (+ ssk_posegraph_optimize)
 */
#ifndef SSK_RBA_H
#define SSK_RBA_H

#include "ssk_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssk_rba ssk_rba;

typedef struct ssk_rba_options {
	/* stereo camera (srba_options.stereo_camera, UTIL.h:366) and camera pose on the robot, row-major [R|t] (EST.cpp:1105-1112) */
	double fx_l, fy_l, cx_l, cy_l, fx_r, fy_r, cx_r, cy_r, baseline;
	double sensor_pose[12];
	/* rba.parameters (EST.cpp:1149-1160; KITTI.ini:104-109) */
	int32_t max_tree_depth, max_optimize_depth, submap_size;
	int32_t min_obs_to_loop_closure;       /* SRBA's default 6 (never set by the reference, SRBA.h:166) */
	int32_t use_robust_kernel, use_robust_kernel_stage1;
	int32_t reserved;                      /* flags.  bit 0: an extra edge to a far local-map centre starts from the previous edge's value, SRBA.h:294
	                                        * as written (default 0: from the composed current estimates -- the literal value diverges) */
	double kernel_param;
} ssk_rba_options;

/* one entry of mySRBA::new_kf_observations_t (EST.cpp:137-160, 671-699): is_fixed = false, is_unknown_with_init_val = true */
typedef struct ssk_rba_obs {
	uint64_t feat_id;
	float left_x, left_y, right_x, right_y;
	double rel_pos[3];                     /* landmark in the new keyframe's robot frame (used when the id is new) */
} ssk_rba_obs;

#define SSK_RBA_MAX_NEW_EDGES 16
typedef struct ssk_rba_kf_info {           /* TNewKeyFrameInfo, the fields the estimator reads (EST.cpp:178-180, 790, 847-849) */
	int32_t kf_id, n_created_edges;
	int32_t created_edge_ids[SSK_RBA_MAX_NEW_EDGES];
	double total_sqr_error_init, total_sqr_error_final;    /* optimize_results */
	double stg1_obs_rmse;                                  /* optimize_results_stg1.obs_rmse */
	int32_t num_observations, num_iters;
} ssk_rba_kf_info;

int ssk_rba_create(ssk_ctx* ctx, const ssk_rba_options* options, ssk_rba** out);
void ssk_rba_destroy(ssk_rba* r);
const char* ssk_rba_last_error(const ssk_rba* r);
int ssk_rba_set_initial_kf_pose(ssk_rba* r, const double pose[12]);
/* Throws nothing: an edge-creation failure ("the node becomes isolated", SRBA.h:304) or a solver failure comes back as a code. */
int ssk_rba_define_kf(ssk_rba* r, const ssk_rba_obs* obs, int n_obs, int run_local_optimization, ssk_rba_kf_info* out);
/* Round 5: n engines (independent streams, all created on the SAME ssk_ctx) advanced in LOCK-STEP: engine i takes one new keyframe with its own
 * observation list obs[i][0 .. n_obs[i]).  What define_new_keyframe (EST.cpp:782-784) does per stream, for a batch of streams: the graph bookkeeping
 * (alloc_keyframe, edge_creation_policy, add_observation) and both LM windows of every engine are built on host threads side by side; stage 1
 * (optimize_edges on the new edges, landmarks held) of ALL engines is ONE ssk_lm_batch (the pose-only kernel), stage 2 (optimize_local_area,
 * depth = max_optimize_depth) another (the node / phase-split kernels).  Every engine ends in the state ssk_rba_define_kf would have left it in.
 * An engine whose bookkeeping fails ("the node becomes isolated", SRBA.h:304) has taken the keyframe and is left out of the solves, the batch goes on
 * for the others; the call then returns the error code with the failed engines' indices in ssk_rba_last_error(r[0]) -- every other engine's record
 * and state are valid, a failed engine's record has kf_id set and zeros otherwise.
 * out: n records.  timing (may be NULL): host-clock milliseconds of the call's parts and the shape of the stage-2 batch. */
typedef struct ssk_rba_batch_timing {
	double bookkeeping_and_windows_ms, stage1_create_ms, stage1_solve_ms, stage2_refresh_ms, stage2_create_ms, stage2_solve_ms, writeback_ms;
	double stage1_symbolic_ms, stage2_symbolic_ms;          /* ssk_lm_batch_symbolic_ms of the two batches (inside the create times) */
	double stage1_mean_obs, stage2_mean_obs, stage2_mean_edges, stage1_mean_iters, stage2_mean_iters;
	int32_t host_threads, n_stage1, n_stage2, stage2_max_edges, stage2_max_obs, reserved;
	char stage1_kernel[96], stage2_kernel[96];
	double teardown_ms, total_ms;                            /* round 6: the windows' host arrays freed; the whole call inside the library */
} ssk_rba_batch_timing;
int ssk_rba_define_kf_batch(ssk_rba* const* r, int n, const ssk_rba_obs* const* obs, const int32_t* n_obs, int run_local_optimization,
                            ssk_rba_kf_info* out, ssk_rba_batch_timing* timing);
/* The observation lists of a batch of streams' new keyframes (EST.cpp:666-704, per stream): match i of stream b (i < n_matches[b]) carries the stereo
 * pixels obs_px[b][i] (a11) and the triangulated position init_pos[b][i] (a10); a match the data association TRACKED (status == SSK_DA_TRACKED and
 * train_idx >= 0) inherits the landmark id of the previous keyframe's match train_idx (ids_prev[b][train_idx]), every other match founds a new landmark
 * with the id (b << 40) + next_id[b]++.  ids_prev == NULL: the streams' first keyframe (every match new).  Outputs: ids_out [n][cap] (-1 beyond the
 * matches: the next call's ids_prev), obs_out [n][cap] records ready for ssk_rba_define_kf_batch (row b, n_matches[b] entries).  Host arrays. */
int ssk_rba_obs_from_da_batch(int n, int cap, const int32_t* n_matches, const int32_t* status, const int32_t* train_idx, const float* obs_px,
                              const double* init_pos, const int64_t* ids_prev, int64_t* next_id, int64_t* ids_out, ssk_rba_obs* obs_out);
int ssk_rba_num_keyframes(const ssk_rba* r);
int ssk_rba_num_edges(const ssk_rba* r);
int ssk_rba_num_landmarks(const ssk_rba* r);
/* the local-map bookkeeping of mySRBA::edge_creation_policy (SRBA.h:67-68, 100-108): the centres keyframe `kf` belongs to, ascending
 * (the last one is getLocalmapCenterID), and the keyframes of the local map centred at `center` (*n = -1 when it is not a centre) */
int ssk_rba_localmap_centers(const ssk_rba* r, int kf, int32_t* out, int cap, int32_t* n);
int ssk_rba_localmap_kfs(const ssk_rba* r, int center, int32_t* out, int cap, int32_t* n);
/* k2k edge: inv_pose = pose of keyframe `from` seen from keyframe `to` */
int ssk_rba_get_edge(const ssk_rba* r, int edge, int32_t* from, int32_t* to, double inv_pose[12]);
/* SSK_ERR_BAD_ARG when the id is unknown */
int ssk_rba_get_landmark(const ssk_rba* r, uint64_t feat_id, int32_t* base_kf, double pos[3]);
/* pose of `target` seen from `root` along the breadth-first tree of depth <= max_depth; *found = 0 when it is farther */
int ssk_rba_spanning_tree_pose(const ssk_rba* r, int root, int target, int max_depth, double pose[12], int32_t* found);
/* the global pose-graph problem (EST.cpp:946-947): node poses [n_keyframes][12] from keyframe `root`, one constraint per edge
 * (edge_ij [n][2] = (to, from), edge_z [n][12] = inv_pose); capacities in, counts out; feed ssk_posegraph_optimize */
int ssk_rba_global_poses(const ssk_rba* r, int root, double* node_poses, int node_cap, int32_t* edge_ij, double* edge_z, int edge_cap,
                         int32_t* n_nodes, int32_t* n_edges);

#ifdef __cplusplus
}
#endif
#endif
