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
} ssk_slam_frame_report;

/* preset: "kitti" = demo/config_imgdir_kitti_srba.ini, "default" = the constructor defaults of UTIL.h:283-321 */
int ssk_slam_create(ssk_ctx* ctx, const char* preset, int width, int height, ssk_slam** out);
void ssk_slam_destroy(ssk_slam* s);
const char* ssk_slam_last_error(const ssk_slam* s);
/* before the first frame.  keys: submap_size max_tree_depth max_optimize_depth lc_distance max_translation max_rotation
 * updated_matches_th up_matches_th_plus vo_id_tracking_th n_feats detect_fast_th residual_th use_initial_pose */
int ssk_slam_set_option(ssk_slam* s, const char* key, double value);
int ssk_slam_load_vocabulary(ssk_slam* s, const char* voc_yml_path);     /* DBoW2 YAML, plain or .gz (BOW.h:59-66) */
int ssk_slam_process_frame(ssk_slam* s, const uint8_t* left, const uint8_t* right, int stride, ssk_slam_frame_report* out);
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
