// srba_stereo_slam_options.h -- the option structs the hot path reads, mirroring the reference's
// TSRBAStereoSLAMOptions / TGeneralOptions (src/srba-stereo-slam_utils.h:87-487).  Field names, meanings and
// defaults are the reference's (constructor at UTIL.h:283-321); kitti() sets demo/config_imgdir_kitti_srba.ini.
// The .ini parsing itself (MRPT CConfigFile) is plumbing outside the hot path and is not reproduced.
#pragma once
#include "ssk_host_types.h"

namespace ssk {

struct TGeneralOptions {                       // UTIL.h:87-216 (subset the facade reads)
	std::string out_dir = ".";                 // [APP_OPTIONS] out_dir
	bool debug = false;                        // write the debug dumps (KF.cpp:60-110, EST.cpp:751-763)
	int verbose_level = 0;
};

struct TSRBAStereoSLAMOptions {                // UTIL.h:221-487
	TStereoCamera stereo_camera;
	Pose3D camera_pose_on_robot_rvt;           // image_pose_on_robot (EST.cpp:1105-1112)
	// detect
	size_t n_levels = 1;                       // fixed by the reference (EST.cpp:1141)
	size_t n_feats = 500;
	int min_ORB_distance = 0;
	int detect_fast_th = 5;
	int adaptive_th_min_matches = 100;
	bool orb_adaptive_fast_th = false;
	bool non_maximal_suppression = false;
	int nms_min_distance = 3;                  // [DETECT] min_distance: window = 2*floor((d-1)/2)+1 (KITTI.ini:9,12)
	// stereo match (stereo-vo [MATCH] section, KITTI.ini:17-24)
	float max_y_diff = 0.f;
	int orb_max_distance = 40;
	float min_disparity = 1.f, max_disparity = 4096.f;
	// inter-frame match / data association
	double ransac_fit_prob = 0.95, max_y_diff_epipolar = 1.5, max_orb_distance_da = 60;
	bool da_filter_by_direction = false, da_filter_by_orb_distance = true;
	bool da_filter_by_fund_matrix = true;      // ssk_da_fundamental_filter (counter-based RANSAC)
	bool da_filter_by_pose_change = true;      // ssk_da_change_in_pose
	double residual_th = 50;
	// stereo-vo [LEAST_SQUARES] section (KITTI.ini:40-50), used by getChangeInPose
	int ls_initial_max_iters = 10, ls_max_iters = 100, ls_max_incr_cost = 3, ls_bad_tracking_th = 5;
	double ls_residual_threshold = 1.0, ls_min_mod_out_vector = 0.001, ls_kernel_param = 3.0;
	bool ls_use_robust_kernel = true;
	bool use_initial_pose = true;
	// SRBA
	double srba_kernel_param = 3.0;
	bool srba_use_robust_kernel = true, srba_use_robust_kernel_stage1 = true;
	size_t srba_submap_size = 15, srba_max_tree_depth = 3, srba_max_optimize_depth = 3;
	// SRBA.h:294 as written gives an extra edge to a FAR local-map centre (a loop-closing link) the value of the previously created edge
	// as its initial pose: off by the distance between the two maps, and this build's LM does not recover from it
	// (tests/test_config3_gpu.py::test_literal_loop_closing_edge_init_diverges: 97 % position error on the revisit sequence against
	// 0.1 %).  false (default): start it from the composition of the current estimates along the graph.  true: as written.
	// Not a reference parameter.
	bool srba_literal_lc_edge_init = false;
	// keyframe creation ([SRBA_KF_CREATION], UTIL.h:255-280,298-316)
	size_t updated_matches_th = 50, up_matches_th_plus = 25, lc_distance = 2;
	double max_rotation = 15., max_translation = 0.30;      // degrees, metres
	int vo_id_tracking_th = 40;
	double query_score_th = 0.04;

	static TSRBAStereoSLAMOptions kitti()      // demo/config_imgdir_kitti_srba.ini
	{
		TSRBAStereoSLAMOptions o;
		o.n_feats = 500; o.detect_fast_th = 20; o.adaptive_th_min_matches = 100; o.orb_adaptive_fast_th = true;   // :112-115
		o.non_maximal_suppression = true; o.nms_min_distance = 5;                                                  // :9-13
		o.max_y_diff = 2.f; o.orb_max_distance = 60;                                                               // :17,24
		o.max_orb_distance_da = 60; o.max_y_diff_epipolar = 2.0; o.ransac_fit_prob = 0.95; o.residual_th = 30;      // :119-122
		o.da_filter_by_direction = true;                                                                           // :124
		o.ls_initial_max_iters = 30; o.ls_max_iters = 30; o.ls_max_incr_cost = 3; o.ls_residual_threshold = 15;     // :41-44
		o.ls_min_mod_out_vector = 0.001; o.ls_bad_tracking_th = 5; o.ls_use_robust_kernel = true; o.ls_kernel_param = 2;   // :45-50
		o.srba_max_tree_depth = 4; o.srba_max_optimize_depth = 5; o.srba_submap_size = 10;                          // :104-106
		o.srba_use_robust_kernel = true; o.srba_use_robust_kernel_stage1 = true; o.srba_kernel_param = 1.5;         // :107-109
		o.updated_matches_th = 60; o.use_initial_pose = true;                                                       // :128,133
		o.max_rotation = 25; o.max_translation = 2.0; o.up_matches_th_plus = 25; o.lc_distance = 10; o.vo_id_tracking_th = 40;   // :126-132
		// camera_pose_on_robot (0,0,0,0,0,0) (+) (0,0,0,-90deg,0,-90deg): camera z forward, x right, y down (EST.cpp:1105-1112)
		const double R[9] = {0, 0, 1, -1, 0, 0, 0, -1, 0}, t[3] = {0, 0, 0};
		o.camera_pose_on_robot_rvt = Pose3D::fromRt(R, t);
		o.max_disparity = 128.f;      // depth >= fx*b/128 = 3 m: without a physical disparity bound a wrong L<->R match makes a
		                              // bogus very near landmark whose leverage in the BA dwarfs the true, far ones
		return o;
	}

	ssk_frontend_params frontend(int fast_th, int orb_th) const
	{
		ssk_frontend_params p;
		p.n_feats = (int32_t)n_feats; p.fast_threshold = fast_th; p.n_levels = (int32_t)n_levels;
		p.nms_radius = non_maximal_suppression ? (nms_min_distance - 1) / 2 : 0;
		p.orb_max_distance = orb_th; p.max_y_diff = max_y_diff;
		p.min_disparity = min_disparity; p.max_disparity = max_disparity; p.harris_k = 0.04f;
		return p;
	}
	ssk_da_params da() const
	{
		ssk_da_params p;
		p.filter_by_direction = da_filter_by_direction; p.filter_by_orb_distance = da_filter_by_orb_distance;
		p.rows_offset = stereo_camera.nrows; p.max_orb_distance_da = (float)max_orb_distance_da;
		return p;
	}
	ssk_fm_params fm(uint64_t seed) const
	{
		ssk_fm_params p{};
		p.threshold = max_y_diff_epipolar; p.confidence = ransac_fit_prob;                 // EST.cpp:2044
		p.norm_cx = 0.5 * stereo_camera.ncols; p.norm_cy = 0.5 * stereo_camera.nrows; p.norm_scale = 2.0 / stereo_camera.ncols;
		p.seed = seed; p.max_iters = 2000; p.min_points = 15;
		return p;
	}
	ssk_cip_params cip() const
	{
		ssk_cip_params p{};
		const TStereoCamera& c = stereo_camera;
		p.fx_l = c.fx_l; p.fy_l = c.fy_l; p.cx_l = c.cx_l; p.cy_l = c.cy_l;
		p.fx_r = c.fx_r; p.fy_r = c.fy_r; p.cx_r = c.cx_r; p.cy_r = c.cy_r; p.baseline = c.baseline;
		p.initial_max_iters = ls_initial_max_iters; p.max_iters = ls_max_iters; p.max_incr_cost = ls_max_incr_cost;
		p.use_robust_kernel = ls_use_robust_kernel; p.residual_threshold = ls_residual_threshold;
		p.min_mod_out_vector = ls_min_mod_out_vector; p.kernel_param = ls_kernel_param; p.bad_tracking_th = ls_bad_tracking_th;
		p.residual_th = residual_th;
		return p;
	}
	ssk_lm_params lm(bool robust) const
	{
		ssk_lm_params p;
		const TStereoCamera& c = stereo_camera;
		p.fx_l = c.fx_l; p.fy_l = c.fy_l; p.cx_l = c.cx_l; p.cy_l = c.cy_l;
		p.fx_r = c.fx_r; p.fy_r = c.fy_r; p.cx_r = c.cx_r; p.cy_r = c.cy_r; p.baseline = c.baseline;
		std::memcpy(p.sensor_pose, camera_pose_on_robot_rvt.m, sizeof p.sensor_pose);
		p.max_iters = 20; p.use_robust_kernel = robust; p.kernel_param = srba_kernel_param;
		p.tau = 1e-3; p.max_lambda = 1e20; p.max_error_per_obs_to_stop = 1e-9; p.converged_rel_decrease = 1e-9;
		return p;
	}
};

}  // namespace ssk
