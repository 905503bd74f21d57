/*
 * ssk_slam.h -- C-ABI of the ESTIMATOR: the calls the reference's main() makes on CSRBAStereoSLAMEstimator
 * (src/srba-stereo-slam_main.cpp:31-60 -> initialize, performStereoSLAM; src/CSRBAStereoSLAMEstimator.h:33-204), for hosts
 * that cannot include the C++ facade.  Exported by libssk_facade.so.  0 = SSK_OK, otherwise an ssk_status code and
 * ssk_slam_last_error has the text (the reference throws / prints and leaves its loop, EST.cpp:792-839, MAIN.cpp:66-75).
 *
 *   estimator.initialize(ini)        (EST.cpp:1099-1339)  ->  ssk_slam_create("kitti", w, h) + ssk_slam_set_option + ssk_slam_load_vocabulary
 *   performStereoSLAM, one frame     (EST.cpp:44-937)     ->  ssk_slam_process_frame
 *   after the loop                   (EST.cpp:945-987)    ->  ssk_slam_optimize_pose_graph
 * The frame-to-frame odometry inside is the build's stand-in for stereo-vo (absent; see host/CSRBAStereoSLAMEstimator.h).
 */
#ifndef SSK_SLAM_H
#define SSK_SLAM_H

#include "ssk_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssk_slam ssk_slam;

typedef struct ssk_slam_frame_report {     /* what one loop iteration decided (the reference prints it) */
	int32_t frame;
	int32_t vo_valid, tracked_from_last_frame, tracked_from_last_kf;
	double vo_pose[12];                    /* this frame's camera seen from the previous frame's */
	int32_t checked_for_kf, inserted_kf, vo_force_new_kf, vo_force_check_tracking, vo_force_check_distance;
	int32_t potential_loop_closure, confirmed_loop_closure, lc_kf_id;      /* lc_kf_id = -1: none */
	int32_t n_query, query_ids[4];         /* place-recognition results (EST.cpp:426), best first */
	double query_scores[4];
	int32_t n_similar, similar_kfs[8];
	int32_t highest_tracked;
	/* filled when inserted_kf: TNewKeyFrameInfo (EST.cpp:178-180, 847-849) */
	int32_t kf_id, n_created_edges;
	double total_sqr_error_init, total_sqr_error_final;
	int32_t n_keyframes, reserved;
	/* the local area define_new_keyframe optimised for this keyframe (stage 2, EST.cpp:782-784): optimised edges and observations of its window */
	int32_t lm_stage2_edges, lm_stage2_obs;
} ssk_slam_frame_report;

typedef struct ssk_slam_timing {           /* cumulative host-clock milliseconds since ssk_slam_create: what an un-batched caller pays */
	double vo_ms;                          /* front-end + frame-to-frame association + incremental pose, every frame (EST.cpp:257-330) */
	double query_ms, da_ms;                /* place-recognition query (EST.cpp:424-427), data association with the similar keyframes (:446-451) */
	double insert_ms;                      /* ids, observation list, rba.define_new_keyframe (EST.cpp:564-850); of which: */
	double lm_window_build_ms, lm_batch_create_ms, lm_solve_ms, lm_readback_ms;   /* per LM solve: window selection, symbolic structure + upload, kernel, state back */
	int32_t frames, kf_frames, lm_solves, lm_iters, lm_max_opt_edges, lm_max_obs;
} ssk_slam_timing;

/* preset: "kitti" = demo/config_imgdir_kitti_srba.ini, "default" = the constructor defaults of UTIL.h:283-321 */
int ssk_slam_create(ssk_ctx* ctx, const char* preset, int width, int height, ssk_slam** out);
void ssk_slam_destroy(ssk_slam* s);
const char* ssk_slam_last_error(const ssk_slam* s);
/* before the first frame.  keys: submap_size max_tree_depth max_optimize_depth lc_distance max_translation max_rotation
 * updated_matches_th up_matches_th_plus vo_id_tracking_th n_feats detect_fast_th residual_th use_initial_pose
 * literal_lc_edge_init (1 = SRBA.h:294 as written; default 0: the literal value diverges) */
int ssk_slam_set_option(ssk_slam* s, const char* key, double value);
/* before the first frame: general_options.debug = true, out_dir = dir ([APP_OPTIONS] debug / out_dir, KITTI.ini:71-72): every inserted
 * keyframe writes info_kf%04d.txt / info_feats_kf%04d.txt (KF.cpp:60-110) and da_info_%04d.txt (EST.cpp:751-763) there */
int ssk_slam_set_debug_dir(ssk_slam* s, const char* dir);
int ssk_slam_load_vocabulary(ssk_slam* s, const char* voc_yml_path);     /* DBoW2 YAML, plain or .gz (BOW.h:59-66) */
int ssk_slam_process_frame(ssk_slam* s, const uint8_t* left, const uint8_t* right, int stride, ssk_slam_frame_report* out);
/* keyframe `kf` as the estimator holds it in m_keyframes (KF.h:99-104): features, L<->R matches and, when match_ids != NULL
 * (out->capacity entries), the landmark id of every match (m_matches_ID) */
int ssk_slam_get_keyframe(const ssk_slam* s, int kf, ssk_kf_out* out, uint64_t* match_ids);
/* the data association of the last frame that was checked for a keyframe (t_kf_da_info, UTIL.h:55-66): entry `which` of the
 * similar keyframes in the report's order.  tracking_first[m] = match index in the other keyframe or -1, tracking_dist[m] its
 * Hamming distance (cap >= the candidate's matches); sorted_pos = its place in the order the ids were taken (EST.cpp:455-470);
 * init_pose = the other camera seen from the candidate's that the change-in-pose filter started from (EST.cpp:1828-1862) */
int ssk_slam_last_da(const ssk_slam* s, int which, int32_t* other_kf, int32_t* tracked, int32_t* sorted_pos,
                     int32_t* tracking_first, double* tracking_dist, int cap, int32_t* n_matches, double* init_pose);
int ssk_slam_get_timing(const ssk_slam* s, ssk_slam_timing* out);
int ssk_slam_num_keyframes(const ssk_slam* s);
int ssk_slam_num_landmarks(const ssk_slam* s);
int ssk_slam_num_edges(const ssk_slam* s);
/* poses [cap][12] of the keyframes in the frame of keyframe 0 after the global pose-graph optimisation; writes the
 * reference's out_kf_poses.txt when path != NULL */
int ssk_slam_optimize_pose_graph(ssk_slam* s, double* poses, int cap, int32_t* n_poses, ssk_pg_result* info, const char* path);

#ifdef __cplusplus
}
#endif
#endif
